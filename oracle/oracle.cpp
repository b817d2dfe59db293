// oracle.cpp — CPU oracle for rpt's hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// A plain IEEE-f64 restatement of the reference algorithm; every function cites the reference
// file:line it follows.  Build with -ffp-contract=off: Rust never contracts a*b+c into an FMA.
//
// Third-party arithmetic that is NOT in /root/reference (Cargo.toml:13-18, no Cargo.lock) is
// restated from the crates' published algorithms and named where it is used:
//   rand 0.8.3       Standard f64, UniformFloat::{new,sample,sample_single}, Bernoulli,
//                    UniformInt::sample
//   rand_distr 0.4.0 UnitDisc, UnitCircle
//   nalgebra-glm 0.10 dot (3-vector: (a+b)+c), normalize (component / norm), gemv column
//                    accumulation, lerp/mix, reflect_vec
// StdRng (ChaCha12, entropy-seeded at renderer.rs:121) is replaced by Philox4x32-10 keyed by
// (seed ; pixel, sample, draw block): the reference stream is not reproducible, only the
// distributions are.  PARITY STATUS: "parity unpinned" vs the reference itself (oracle.h).

#include "oracle.h"

// exp/ln/atan/sin_cos/acos/atan2: the reference calls the platform libm through Rust's std.
// Default build: the fdlibm restatement of include/rpt_math.h (pure IEEE arithmetic, so the
// gfx950 kernels reproduce it bit for bit).  -DORACLE_SYSTEM_LIBM: the host's glibc instead
// (liboracle_sysm.so) — tests/test_rpt_math.py shows the two builds agree.
#include "../include/rpt_math.h"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double FRAC_1_PI = 0.318309886183790671537767526745028724;
constexpr double TAU = 6.28318530717958647692528676655900577;
constexpr double EPSILON = 1e-12;        // renderer.rs:14
constexpr double FIREFLY_CLAMP = 100.0;  // renderer.rs:15
constexpr double SCORE_THRESHOLD = 0.85; // kdtree.rs:6

// Rust f64::min / f64::max ignore a NaN operand == C fmin / fmax.
inline double rmin(double a, double b) { return std::fmin(a, b); }
inline double rmax(double a, double b) { return std::fmax(a, b); }
inline bool sign_negative(double x) { return std::signbit(x); }  // f64::is_sign_negative
inline bool sign_positive(double x) { return !std::signbit(x); } // f64::is_sign_positive
inline double signum(double x) {                                 // f64::signum
  if (std::isnan(x)) return x;
  return std::signbit(x) ? -1.0 : 1.0;
}
inline bool is_normal(double x) { return std::isnormal(x); } // f64::is_normal

#ifdef ORACLE_SYSTEM_LIBM
inline double m_exp(double x) { return std::exp(x); }
inline double m_log(double x) { return std::log(x); }
inline double m_atan(double x) { return std::atan(x); }
inline void m_sincos(double x, double& s, double& c) { s = std::sin(x); c = std::cos(x); }
inline double m_acos(double x) { return std::acos(x); }
inline double m_atan2(double y, double x) { return std::atan2(y, x); }
#else
inline double m_exp(double x) { return rpt_exp(x); }
inline double m_log(double x) { return rpt_log(x); }
inline double m_atan(double x) { return rpt_atan(x); }
inline void m_sincos(double x, double& s, double& c) { rpt_sincos_pio2(x, &s, &c); }
inline double m_acos(double x) { return rpt_acos(x); }
inline double m_atan2(double y, double x) { return rpt_atan2(y, x); }
#endif

// ------------------------------------------------------------------ vectors (nalgebra)
struct V3 {
  double x, y, z;
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; } // component_mul
// nalgebra dot for a 3-vector: a + b + c with a,b,c the three products, left to right
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline double length2(V3 a) { return dot(a, a); }
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a / length(a); } // self / self.norm()
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 from(const double* p) { return {p[0], p[1], p[2]}; }
inline void to(V3 a, double* p) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

// column-major 4x4 * (v, w): gemv accumulates column by column, left to right
inline V3 mat4_mul(const double* m, V3 v, double w) {
  V3 r;
  r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * w;
  r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * w;
  r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * w;
  return r;
}
inline V3 mat3_mul(const double* m, V3 v) {
  V3 r;
  r.x = (m[0] * v.x + m[3] * v.y) + m[6] * v.z;
  r.y = (m[1] * v.x + m[4] * v.y) + m[7] * v.z;
  r.z = (m[2] * v.x + m[5] * v.y) + m[8] * v.z;
  return r;
}

// ------------------------------------------------------------------ counters
struct Counters : OracleCounters {
  Counters() { std::memset(static_cast<OracleCounters*>(this), 0, sizeof(OracleCounters)); }
  void add(const OracleCounters& o) {
    uint64_t* a = reinterpret_cast<uint64_t*>(static_cast<OracleCounters*>(this));
    const uint64_t* b = reinterpret_cast<const uint64_t*>(&o);
    for (size_t i = 0; i < sizeof(OracleCounters) / sizeof(uint64_t); i++) a[i] += b[i];
  }
};
thread_local Counters* tl_cnt = nullptr;
thread_local bool tl_shadow = false; // geometry visited on behalf of a shadow ray (renderer.rs:191)
#ifdef ORACLE_NO_COUNTERS
// the CPU-baseline builds (liboracle_fast.so, liboracle_native.so): no visit counting in the hot loops, which is
// what the reference's own binary looks like; counters requested from such a build read zero
#define COUNT(field) do { } while (0)
#define COUNT_GEO(field) do { } while (0)
#else
#define COUNT(field)            \
  do {                          \
    if (tl_cnt) tl_cnt->field++; \
  } while (0)
// geometry counters are kept separately for closest-hit rays and for shadow rays
#define COUNT_GEO(field)                                   \
  do {                                                     \
    if (tl_cnt) {                                          \
      if (tl_shadow) tl_cnt->field##_sh++;                 \
      else tl_cnt->field++;                                \
    }                                                      \
  } while (0)
#endif

// ------------------------------------------------------------------ RNG
// Philox4x32-10 (Salmon et al., SC'11), the Random123 reference constants.
inline void philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
  uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
  uint32_t k0 = k_in[0], k1 = k_in[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The stream of one camera path: key = seed, counter = (pixel, sample lo, sample hi, block);
// each block yields two u64 draws (words 0|1<<32 then 2|3<<32).
struct Rng {
  uint32_t key[2];
  uint32_t pixel;
  uint32_t slo, shi;
  uint32_t draw = 0; // number of u64 draws consumed so far
  Rng(uint64_t seed, uint32_t pixel_, uint64_t sample) {
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    pixel = pixel_;
    slo = (uint32_t)sample;
    shi = (uint32_t)(sample >> 32);
  }
  uint64_t next_u64() {
    COUNT(rng_draws);
    uint32_t ctr[4] = {pixel, slo, shi, draw >> 1};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    uint64_t r = (draw & 1) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
    draw++;
    return r;
  }
  // rand 0.8 Standard for f64: 53 random bits, [0,1)
  double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
  // rand 0.8 into_float_with_exponent(0) - 1.0 on the top 52 bits: [0,1) with 52 bits
  double gen_u52() {
    uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
    double value1_2;
    std::memcpy(&value1_2, &bits, 8);
    return value1_2 - 1.0;
  }
  // rand 0.8 UniformFloat::<f64>::sample_single (Rng::gen_range(lo..hi)): redraw if the
  // rounded result reaches `high`
  double gen_range(double low, double high) {
    double scale = high - low;
    for (;;) {
      double res = gen_u52() * scale + low;
      if (res < high) return res;
    }
  }
  // rand 0.8 Bernoulli: p == 1 always true without a draw, else u64 < p * 2^64
  bool gen_bool(double p) {
    if (p == 1.0) return true;
    uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
    return next_u64() < p_int;
  }
  // rand 0.8 Standard for bool: (next_u32() as i32) < 0; next_u32 here = low word of one u64 draw
  bool gen_bool_std() { return (int32_t)(uint32_t)next_u64() < 0; }
  // rand 0.8 UniformInt::<usize>::sample for Uniform::from(0..n): widening multiply + zone
  uint64_t gen_index(uint64_t n) {
    uint64_t range = n;
    uint64_t ints_to_reject = (UINT64_MAX - range + 1) % range;
    uint64_t zone = UINT64_MAX - ints_to_reject;
    for (;;) {
      uint64_t v = next_u64();
      __uint128_t m = (__uint128_t)v * range;
      uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
      if (lo <= zone) return hi;
    }
  }
  // Uniform::new(-1.0, 1.0).sample: value0_1 * scale + low with scale = 2 (no shrink needed:
  // 2*(1-2^-52)-1 < 1)
  double gen_pm1() { return gen_u52() * 2.0 + -1.0; }
  // rand_distr 0.4 UnitDisc: rejection in [-1,1)^2, accept x^2+y^2 <= 1
  void unit_disc(double& x, double& y) {
    for (;;) {
      x = gen_pm1();
      y = gen_pm1();
      if (x * x + y * y <= 1.0) return;
    }
  }
  // rand_distr 0.4 UnitCircle: von Neumann, accept sum < 1, ((x^2-y^2)/s, 2xy/s)
  void unit_circle(double& ox, double& oy) {
    double x1, x2, sum;
    for (;;) {
      x1 = gen_pm1();
      x2 = gen_pm1();
      sum = x1 * x1 + x2 * x2;
      if (sum < 1.0) break;
    }
    double diff = x1 * x1 - x2 * x2;
    ox = diff / sum;
    oy = 2.0 * x1 * x2 / sum;
  }
};

// ------------------------------------------------------------------ Ray / HitRecord
struct Ray { // shape.rs:49-55
  V3 origin, dir;
  V3 at(double time) const { return origin + time * dir; } // shape.rs:59-61
};
struct HitRecord { // shape.rs:75-90
  double time = INF;
  V3 normal = {0, 0, 0};
};
struct Sample { V3 v, n; double p; };

// ------------------------------------------------------------------ BoundingBox
struct BBox { // kdtree.rs:28-42
  V3 p_min = {INF, INF, INF};
  V3 p_max = {-INF, -INF, -INF};
  BBox merge(const BBox& o) const { // kdtree.rs:46-51
    return {{rmin(p_min.x, o.p_min.x), rmin(p_min.y, o.p_min.y), rmin(p_min.z, o.p_min.z)},
            {rmax(p_max.x, o.p_max.x), rmax(p_max.y, o.p_max.y), rmax(p_max.z, o.p_max.z)}};
  }
  void intersect(const Ray& r, double& t0, double& t1) const { // kdtree.rs:54-68
    double x1 = (p_min.x - r.origin.x) / r.dir.x;
    double x2 = (p_max.x - r.origin.x) / r.dir.x;
    double xa = rmin(x1, x2), xb = rmax(x1, x2);
    double y1 = (p_min.y - r.origin.y) / r.dir.y;
    double y2 = (p_max.y - r.origin.y) / r.dir.y;
    double ya = rmin(y1, y2), yb = rmax(y1, y2);
    double z1 = (p_min.z - r.origin.z) / r.dir.z;
    double z2 = (p_max.z - r.origin.z) / r.dir.z;
    double za = rmin(z1, z2), zb = rmax(z1, z2);
    t0 = rmax(rmax(xa, ya), za);
    t1 = rmin(rmin(xb, yb), zb);
  }
  void split(int axis, double value, BBox& lo, BBox& hi) const { // kdtree.rs:71-86
    lo = *this;
    hi = *this;
    if (axis == 0) { lo.p_max.x = value; hi.p_min.x = value; }
    else if (axis == 1) { lo.p_max.y = value; hi.p_min.y = value; }
    else { lo.p_max.z = value; hi.p_min.z = value; }
  }
};

// ------------------------------------------------------------------ Shape / Bounded traits
struct Shape { // shape.rs:18-25 ; Bounded kdtree.rs:9-12
  virtual ~Shape() {}
  virtual bool intersect(const Ray& ray, double t_min, HitRecord& rec) const = 0;
  virtual Sample sample(V3 target, Rng& rng) const = 0;
  virtual bool bounded() const { return true; }
  virtual BBox bounding_box() const = 0;
};

struct Sphere : Shape { // sphere.rs
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // :13-45
    COUNT_GEO(n_sphere);
    double a = length2(ray.dir);
    double b = dot(ray.dir, ray.origin);
    double c = length2(ray.origin) - 1.0;
    double d = b * b - a * c;
    if (sign_negative(d)) return false;
    d = std::sqrt(d);
    double t;
    double t_minus = (-b - d) / a;
    if (t_minus < t_min) {
      double t_plus = (-b + d) / a;
      if (t_plus < t_min) return false;
      t = t_plus;
    } else {
      t = t_minus;
    }
    if (t < rec.time) {
      rec.time = t;
      rec.normal = normalize(ray.at(t));
      return true;
    }
    return false;
  }
  Sample sample(V3 target, Rng& rng) const override { // :52-64
    double x, y;
    rng.unit_disc(x, y);
    double z = std::sqrt(1.0 - x * x - y * y);
    V3 n = normalize(target);
    V3 n1 = is_normal(n.x) ? normalize(v3(n.y, -n.x, 0.0)) : normalize(v3(0.0, -n.z, n.y));
    V3 n2 = cross(n1, n);
    V3 p = x * n1 + y * n2 + z * n;
    return {p, p, z * FRAC_1_PI};
  }
  BBox bounding_box() const override { return {{-1, -1, -1}, {1, 1, 1}}; } // :66-73
};

struct Plane : Shape { // plane.rs
  V3 normal;
  double value;
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // :17-32
    COUNT_GEO(n_plane);
    double cosine = dot(normal, ray.dir);
    if (std::fabs(cosine) < 1e-8) return false;
    double time = (value - dot(normal, ray.origin)) / cosine;
    if (time >= t_min && time < rec.time) {
      rec.time = time;
      rec.normal = (-normalize(normal)) * signum(cosine);
      return true;
    }
    return false;
  }
  Sample sample(V3, Rng&) const override { // :34-36 unimplemented!()
    std::abort();
  }
  bool bounded() const override { return false; }
  BBox bounding_box() const override { std::abort(); }
};

struct Cube : Shape { // cube.rs
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // :20-72
    COUNT_GEO(n_cube);
    double t1[3], t2[3];
    V3 n1[3], n2[3];
    for (int dim = 0; dim < 3; dim++) { // compute_interval :21-33
      double x1 = (-0.5 - ray.origin[dim]) / ray.dir[dim];
      double x2 = (0.5 - ray.origin[dim]) / ray.dir[dim];
      V3 x1n = {0, 0, 0}, x2n = {0, 0, 0};
      (dim == 0 ? x1n.x : dim == 1 ? x1n.y : x1n.z) = -1.0;
      (dim == 0 ? x2n.x : dim == 1 ? x2n.y : x2n.z) = 1.0;
      if (x1 > x2) {
        std::swap(x1, x2);
        std::swap(x1n, x2n);
      }
      t1[dim] = x1; t2[dim] = x2; n1[dim] = x1n; n2[dim] = x2n;
    }
    double start, end;
    V3 start_normal, end_normal;
    if (t1[0] > t1[1] && t1[0] > t1[2]) { start = t1[0]; start_normal = n1[0]; } // :38-46
    else if (t1[1] > t1[2]) { start = t1[1]; start_normal = n1[1]; }
    else { start = t1[2]; start_normal = n1[2]; }
    if (t2[0] < t2[1] && t2[0] < t2[2]) { end = t2[0]; end_normal = n2[0]; } // :47-55
    else if (t2[1] < t2[2]) { end = t2[1]; end_normal = n2[1]; }
    else { end = t2[2]; end_normal = n2[2]; }
    if (start > end || end < t_min) return false; // :57-59
    double time;
    V3 normal;
    if (start < t_min) { time = end; normal = end_normal; }
    else { time = start; normal = start_normal; }
    if (time < rec.time) {
      rec.time = time;
      rec.normal = normal;
      return true;
    }
    return false;
  }
  Sample sample(V3, Rng& rng) const override { // :74-87
    double a = rng.gen_f64() - 0.5;
    double b = rng.gen_f64() - 0.5;
    V3 v, n;
    switch (rng.gen_index(6)) {
      case 0: v = v3(a, b, 0.5); n = v3(0, 0, 1); break;
      case 1: v = v3(a, b, -0.5); n = v3(0, 0, -1); break;
      case 2: v = v3(a, 0.5, b); n = v3(0, 1, 0); break;
      case 3: v = v3(a, -0.5, b); n = v3(0, -1, 0); break;
      case 4: v = v3(0.5, a, b); n = v3(1, 0, 0); break;
      default: v = v3(-0.5, a, b); n = v3(-1, 0, 0); break;
    }
    return {v, n, 1.0 / 6.0};
  }
  BBox bounding_box() const override { return {{-0.5, -0.5, -0.5}, {0.5, 0.5, 0.5}}; } // :10-17
};

struct Monomial : Shape { // monomial_surface.rs (exp = 4 only, :10)
  double height = 1.0, exp = 4.0;
  BBox bounding_box() const override { return {{-1.0, 0.0, -1.0}, {1.0, height, 1.0}}; } // :183-190
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // :21-106
    COUNT_GEO(n_monomial);
    double b_min, b_max;
    bounding_box().intersect(ray, b_min, b_max);
    if (rmax(b_min, t_min) > rmin(b_max, rec.time)) return false;
    const V3 o = ray.origin, d = ray.dir;
    auto dist = [&](double t) { // :26-31 ; powi(2) = s * s
      double x = o.x + t * d.x, y = o.y + t * d.y, z = o.z + t * d.z;
      double s = x * x + z * z;
      return y - height * (s * s);
    };
    double coef0 = o.x * o.x + o.z * o.z;
    double coef1 = 2. * (o.x * d.x + o.z * d.z);
    double coef2 = d.x * d.x + d.z * d.z;
    auto deriv = [&](double t) { // :35-41 ; left-to-right as written
      double dy = 2. * coef0 * coef1 + 2. * t * (coef1 * coef1 + 2. * coef0 * coef2)
                  + 3. * (t * t) * 2. * coef1 * coef2 + 4. * (t * (t * t)) * coef2 * coef2;
      return d.y - height * dy;
    };
    auto deriv2 = [&](double t) { // :42-47
      double dy = 2. * (coef1 * coef1 + 2. * coef0 * coef2) + 3. * 2. * t * 2. * coef1 * coef2
                  + 4. * 3. * (t * t) * coef2 * coef2;
      return -height * dy;
    };
    double t_max;
    bool maximize = dist(t_min) < 0.0;
    if (maximize) { // Newton on the distance from inside/below, :50-67
      double cur_x = (b_min + b_max) / 2.;
      for (int i = 0; i < 10; i++) {
        double f = dist(cur_x);
        if (f > 0.) break;
        double der = deriv(cur_x), der2 = deriv2(cur_x);
        cur_x -= der / der2;
      }
      t_max = cur_x; // (:61-63 only prints a diagnostic)
      if (t_max < t_min) return false;
    } else {
      t_max = 10000.;
    }
    if ((dist(t_min) < 0.0) == (dist(t_max) < 0.0)) return false;
    double l = t_min, r = t_max;
    for (int i = 0; i < 60; i++) { // :74-81
      double m = (l + r) / 2.0;
      if ((dist(m) >= 0.0) == maximize) r = m;
      else l = m;
    }
    if (r > rec.time) return false;
    V3 pos = ray.at(r);
    if (pos.x * pos.x + pos.z * pos.z > 1.0) return false; // :86-89
    rec.time = r;
    double s = pos.x * pos.x + pos.z * pos.z;
    rec.normal = normalize(v3(height * 4.0 * pos.x * s, -1.0, height * 4.0 * pos.z * s)); // :92-96
    if (dot(rec.normal, ray.dir) > 0.0) rec.normal = -rec.normal; // two-sided, :99-101
    return true;
  }
  Sample sample(V3, Rng& rng) const override { // :108-123
    double x, z;
    rng.unit_circle(x, z);
    // (x*x + z*z).powf(exp / 2.) with exp = 4: pow(s, 2.0), restated as s * s (the correctly rounded
    // square; a libm pow may differ from it by one ulp on rare arguments)
    double s = x * x + z * z;
    V3 pos = v3(x, height * (s * s), z);
    V3 normal = normalize(v3(height * 4. * pos.x * (pos.x * pos.x + pos.z * pos.z), -1.,
                             height * 4. * pos.z * (pos.x * pos.x + pos.z * pos.z)));
    const double AREA = 6.3406654362;
    if (rng.gen_bool_std()) normal = -normal;
    return {pos, normal, 1. / (2. * AREA)};
  }
  V3 closest_point(V3 point, int steps) const { // :126-152 (steps = 100), :154-181 (steps = 10000)
    if (length(point) < 1e-12) return point;
    double px = std::hypot(point.x, point.z), py = point.y;
    double best = 1e18, best_x = -1.;
    for (int i = -steps; i <= steps; i++) {
      double xf = (double)i / (double)steps;
      double x4 = (xf * xf) * (xf * xf); // powi(4)
      double dx = px - xf, dy = py - height * x4;
      double dist2 = dx * dx + dy * dy;
      if (dist2 < best) { best = dist2; best_x = xf; }
    }
    double n = std::sqrt(point.x * point.x + point.z * point.z);
    double qx = best_x * (point.x / n), qz = best_x * (point.z / n);
    double r2 = qx * qx + qz * qz;
    return v3(qx, height * (r2 * r2), qz);
  }
};

struct Triangle : Shape { // mesh.rs:8-22
  V3 v1, v2, v3_, n1, n2, n3;
  BBox bounding_box() const override { // mesh.rs:40-45
    return {{rmin(rmin(v1.x, v2.x), v3_.x), rmin(rmin(v1.y, v2.y), v3_.y), rmin(rmin(v1.z, v2.z), v3_.z)},
            {rmax(rmax(v1.x, v2.x), v3_.x), rmax(rmax(v1.y, v2.y), v3_.y), rmax(rmax(v1.z, v2.z), v3_.z)}};
  }
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // mesh.rs:49-82
    COUNT_GEO(n_tri);
    V3 d0 = v2 - v1, d1 = v3_ - v1;
    V3 plane_normal = normalize(cross(d0, d1));
    double cosine = dot(plane_normal, ray.dir);
    if (std::fabs(cosine) < 1e-8) return false;
    double time = dot(plane_normal, v1 - ray.origin) / cosine;
    if (time < t_min || time >= rec.time) return false;
    V3 d2 = ray.at(time) - v1;
    double d00 = dot(d0, d0);
    double d01 = dot(d0, d1);
    double d11 = dot(d1, d1);
    double d20 = dot(d2, d0);
    double d21 = dot(d2, d1);
    double denom = d00 * d11 - d01 * d01;
    double v = (d11 * d20 - d01 * d21) / denom;
    double w = (d00 * d21 - d01 * d20) / denom;
    double u = 1.0 - v - w;
    if (u >= 0.0 && v >= 0.0 && w >= 0.0) {
      rec.time = time;
      rec.normal = normalize(u * n1 + v * n2 + w * n3);
      return true;
    }
    return false;
  }
  Sample sample(V3, Rng& rng) const override { // mesh.rs:84-98
    double u = rng.gen_f64();
    double v = rng.gen_f64();
    while (u + v > 1.0) {
      u = rng.gen_f64();
      v = rng.gen_f64();
    }
    double w = 1.0 - u - v;
    double area = 0.5 * length(cross(v2 - v1, v3_ - v1));
    return {u * v1 + v * v2 + w * v3_, normalize(u * n1 + v * n2 + w * n3), 1.0 / area};
  }
};

struct Transformed : Shape { // shape.rs:101-176
  std::unique_ptr<Shape> shape;
  RptTransform xf;
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // :128-137
    COUNT_GEO(n_inst);
    Ray local; // Ray::apply_transform shape.rs:64-71 (direction NOT renormalised)
    local.origin = mat4_mul(xf.inverse_transform, ray.origin, 1.0);
    local.dir = mat4_mul(xf.inverse_transform, ray.dir, 0.0);
    if (shape->intersect(local, t_min, rec)) {
      rec.normal = normalize(mat3_mul(xf.normal_transform, rec.normal));
      return true;
    }
    return false;
  }
  Sample sample(V3 target, Rng& rng) const override { // :139-151
    V3 t = mat4_mul(xf.inverse_transform, target, 1.0);
    Sample s = shape->sample(t, rng);
    V3 new_normal = normalize(mat3_mul(xf.normal_transform, s.n));
    double parallelepiped_height = dot(mat3_mul(xf.linear, s.n), new_normal);
    double parallelepiped_base = xf.scale / parallelepiped_height;
    return {mat4_mul(xf.transform, s.v, 1.0), new_normal, s.p / parallelepiped_base};
  }
  bool bounded() const override { return shape->bounded(); }
  BBox bounding_box() const override { // :153-176
    BBox b = shape->bounding_box();
    V3 c[8];
    int k = 0;
    for (int ix = 0; ix < 2; ix++)
      for (int iy = 0; iy < 2; iy++)
        for (int iz = 0; iz < 2; iz++)
          c[k++] = mat4_mul(xf.transform,
                            v3(ix ? b.p_max.x : b.p_min.x, iy ? b.p_max.y : b.p_min.y,
                               iz ? b.p_max.z : b.p_min.z), 1.0);
    BBox r;
    for (int i = 0; i < 8; i++) r = r.merge({c[i], c[i]});
    return r;
  }
};

// ------------------------------------------------------------------ KdTree
struct KdNode { // kdtree.rs:227-233
  int axis = 3; // 0,1,2 = SplitX/Y/Z ; 3 = Leaf
  double value = 0;
  std::unique_ptr<KdNode> left, right;
  std::vector<size_t> indices;
};

double median(const std::vector<double>& a) { // kdtree.rs:347-355
  size_t n = a.size();
  if (n % 2 == 1) return a[n / 2];
  size_t mid = n / 2;
  return (a[mid] + a[mid - 1]) / 2.0;
}

std::unique_ptr<KdNode> construct(const std::vector<BBox>& all, std::vector<size_t> indices) {
  // kdtree.rs:235-345
  auto node = std::make_unique<KdNode>();
  if (indices.size() < 16) {
    node->indices = std::move(indices);
    return node;
  }
  std::vector<double> xs, ys, zs;
  std::vector<BBox> bboxs;
  for (size_t index : indices) {
    const BBox& b = all[index];
    xs.push_back(b.p_min.x); xs.push_back(b.p_max.x);
    ys.push_back(b.p_min.y); ys.push_back(b.p_max.y);
    zs.push_back(b.p_min.z); zs.push_back(b.p_max.z);
    bboxs.push_back(b);
  }
  // kdtree.rs:251-254: `sort_by(partial_cmp)` is a STABLE sort under which -0.0 and +0.0 are equal — entries that
  // compare equal keep the order they were pushed in (per index: p_min, p_max).  It matters in one case: zeros of both
  // signs around the middle of the array decide the SIGN of a zero median ((-0 + -0) / 2 = -0, every other pair +0).
  std::stable_sort(xs.begin(), xs.end());
  std::stable_sort(ys.begin(), ys.end());
  std::stable_sort(zs.begin(), zs.end());
  double m[3] = {median(xs), median(ys), median(zs)};
  auto partition_score = [&](int dim, double value) {
    size_t left = 0, right = 0;
    for (const BBox& b : bboxs) {
      if (b.p_min[dim] <= value) left++;
      if (b.p_max[dim] >= value) right++;
    }
    return std::max(left, right);
  };
  size_t s[3] = {partition_score(0, m[0]), partition_score(1, m[1]), partition_score(2, m[2])};
  size_t threshold = (size_t)((double)indices.size() * SCORE_THRESHOLD);
  if (std::min(std::min(s[0], s[1]), s[2]) >= threshold) {
    node->indices = std::move(indices);
    return node;
  }
  int split_dir = -1;
  BBox bounds;
  for (const BBox& b : bboxs) bounds = bounds.merge(b);
  V3 extent = bounds.p_max - bounds.p_min;
  if (extent.x > extent.y && extent.x > extent.z) {
    if (s[0] < threshold) split_dir = 0;
  } else if (extent.y > extent.z) {
    if (s[1] < threshold) split_dir = 1;
  } else if (s[2] < threshold) {
    split_dir = 2;
  }
  if (split_dir == -1) {
    if (s[0] < s[1] && s[0] < s[2]) split_dir = 0;
    else if (s[1] < s[2]) split_dir = 1;
    else split_dir = 2;
  }
  std::vector<size_t> left, right;
  for (size_t i = 0; i < indices.size(); i++) {
    if (bboxs[i].p_min[split_dir] <= m[split_dir]) left.push_back(indices[i]);
    if (bboxs[i].p_max[split_dir] >= m[split_dir]) right.push_back(indices[i]);
  }
  node->axis = split_dir;
  node->value = m[split_dir];
  node->left = construct(all, std::move(left));
  node->right = construct(all, std::move(right));
  return node;
}

struct KdTree : Shape { // kdtree.rs:100-144
  std::unique_ptr<KdNode> root;
  std::vector<std::unique_ptr<Shape>> objects;
  BBox bounds;
  void build() { // KdTree::new kdtree.rs:108-119
    std::vector<BBox> boxes;
    std::vector<size_t> idx;
    for (size_t i = 0; i < objects.size(); i++) {
      boxes.push_back(objects[i]->bounding_box());
      idx.push_back(i);
    }
    bounds = BBox();
    for (const BBox& b : boxes) bounds = bounds.merge(b);
    root = construct(boxes, std::move(idx));
  }
  BBox bounding_box() const override { return bounds; } // :122-126
  bool intersect(const Ray& ray, double t_min, HitRecord& rec) const override { // :129-136
    COUNT_GEO(n_root);
    double b_min, b_max;
    bounds.intersect(ray, b_min, b_max);
    if (rmax(b_min, t_min) > rmin(b_max, rec.time)) return false;
    return intersect_subtree(*root, bounds, ray, t_min, rec);
  }
  Sample sample(V3 target, Rng& rng) const override { // :138-143
    size_t num = objects.size();
    size_t index = rng.gen_index(num);
    Sample s = objects[index]->sample(target, rng);
    return {s.v, s.n, s.p / (double)num};
  }
  bool intersect_subtree(const KdNode& node, const BBox& bbox, const Ray& ray, double t_min,
                         HitRecord& rec) const { // :151-223
    double b_min, b_max;
    bbox.intersect(ray, b_min, b_max);
    if (node.axis == 3) {
      COUNT_GEO(n_leaf);
      bool result = false;
      for (size_t index : node.indices) {
        COUNT_GEO(n_ref);
        if (objects[index]->intersect(ray, t_min, rec)) result = true;
      }
      return result;
    }
    COUNT_GEO(n_inner);
    double o = ray.origin[node.axis], d = ray.dir[node.axis];
    double t_split = (node.value - o) / d;
    bool left_first = (o < node.value) || (o == node.value && d <= 0.0);
    BBox bl, br;
    bbox.split(node.axis, node.value, bl, br);
    const KdNode* first = left_first ? node.left.get() : node.right.get();
    const KdNode* second = left_first ? node.right.get() : node.left.get();
    const BBox& b0 = left_first ? bl : br;
    const BBox& b1 = left_first ? br : bl;
    if (t_split > rmin(b_max, rec.time) || t_split <= 0.0) {
      return intersect_subtree(*first, b0, ray, t_min, rec);
    } else if (t_split < rmax(b_min, t_min)) {
      return intersect_subtree(*second, b1, ray, t_min, rec);
    } else {
      bool h1 = intersect_subtree(*first, b0, ray, t_min, rec);
      if (h1 && rec.time < t_split) return true;
      bool h2 = intersect_subtree(*second, b1, ray, t_split, rec);
      return h1 || h2;
    }
  }
};

// ------------------------------------------------------------------ shape factory
std::unique_ptr<Shape> make_shape(const RptShape& d, int depth = 0) {
  std::unique_ptr<Shape> inner;
  switch (d.kind) {
    case RPT_SHAPE_SPHERE: inner = std::make_unique<Sphere>(); break;
    case RPT_SHAPE_CUBE: inner = std::make_unique<Cube>(); break;
    case RPT_SHAPE_PLANE: {
      auto p = std::make_unique<Plane>();
      p->normal = from(d.plane_normal);
      p->value = d.plane_value;
      inner = std::move(p);
      break;
    }
    case RPT_SHAPE_MONOMIAL: {
      if (d.monomial_exp != 4.0) return nullptr; // monomial_surface.rs:10
      auto m = std::make_unique<Monomial>();
      m->height = d.monomial_height;
      m->exp = d.monomial_exp;
      inner = std::move(m);
      break;
    }
    case RPT_SHAPE_MESH: {
      auto t = std::make_unique<KdTree>();
      for (uint64_t i = 0; i < d.num_triangles; i++) {
        auto tri = std::make_unique<Triangle>();
        const RptTriangle& s = d.triangles[i];
        tri->v1 = from(s.v1); tri->v2 = from(s.v2); tri->v3_ = from(s.v3);
        tri->n1 = from(s.n1); tri->n2 = from(s.n2); tri->n3 = from(s.n3);
        t->objects.push_back(std::move(tri));
      }
      t->build();
      inner = std::move(t);
      break;
    }
    case RPT_SHAPE_GROUP: {
      auto t = std::make_unique<KdTree>();
      for (uint64_t i = 0; i < d.num_children; i++) {
        auto c = make_shape(d.children[i], depth + 1);
        if (!c || !c->bounded()) return nullptr;
        t->objects.push_back(std::move(c));
      }
      t->build();
      inner = std::move(t);
      break;
    }
    default: return nullptr;
  }
  if (d.transformed) {
    auto t = std::make_unique<Transformed>();
    t->shape = std::move(inner);
    t->xf = d.xf;
    return t;
  }
  return inner;
}

// ------------------------------------------------------------------ Material
inline double pow2(double x) { return x * x; }           // powi(2)
inline double pow3(double x) { return x * (x * x); }     // powi(3): compiler-rt __powidf2
inline double pow5(double x) { double x2 = x * x; return x * (x2 * x2); } // powi(5)
inline V3 lerp(V3 a, V3 b, double t) { return a * (1.0 - t) + b * t; }   // glm::lerp / mix

V3 bsdf(const RptMaterial& m, V3 n, V3 wo, V3 wi) { // material.rs:125-210
  V3 color = from(m.color);
  double n_dot_wi = dot(n, wi);
  double n_dot_wo = dot(n, wo);
  bool wi_outside = sign_positive(n_dot_wi);
  bool wo_outside = sign_positive(n_dot_wo);
  if (!m.transparent && (!wi_outside || !wo_outside)) return {0, 0, 0};
  const V3 one = {1, 1, 1};
  if (wi_outside == wo_outside) {
    V3 h = normalize(wi + wo);
    double wo_dot_h = dot(wo, h);
    double n_dot_h = dot(n, h);
    double nh2 = pow2(n_dot_h);
    double m2 = m.roughness * m.roughness;
    double d = m_exp((nh2 - 1.0) / (m2 * nh2)) / (m2 * PI * nh2 * nh2);
    V3 f;
    if (!wi_outside && std::sqrt(1.0 - wo_dot_h * wo_dot_h) * m.index > 1.0) {
      f = one;
    } else {
      double f0s = pow2((m.index - 1.0) / (m.index + 1.0));
      V3 f0 = lerp(v3(f0s, f0s, f0s), color, m.metallic);
      f = f0 + (one - f0) * pow5(1.0 - wo_dot_h);
    }
    double g = rmin(n_dot_wi * n_dot_h, n_dot_wo * n_dot_h);
    g = (2.0 * g) / wo_dot_h;
    g = rmin(g, 1.0);
    V3 specular = d * f * g / (4.0 * n_dot_wo * n_dot_wi);
    if (m.transparent) return specular;
    V3 diffuse = cmul(one - f, color) / PI;
    return specular + diffuse;
  } else {
    double eta_t = wo_outside ? m.index : 1.0 / m.index;
    V3 h = normalize(wi * eta_t + wo);
    double wi_dot_h = dot(wi, h);
    double wo_dot_h = dot(wo, h);
    double n_dot_h = dot(n, h);
    double nh2 = pow2(n_dot_h);
    double m2 = m.roughness * m.roughness;
    double d = m_exp((nh2 - 1.0) / (m2 * nh2)) / (m2 * PI * nh2 * nh2);
    double f0s = pow2((m.index - 1.0) / (m.index + 1.0));
    V3 f0 = lerp(v3(f0s, f0s, f0s), color, m.metallic);
    V3 f = f0 + (one - f0) * pow5(1.0 - std::fabs(wi_dot_h));
    double g = rmin(std::fabs(n_dot_wi * n_dot_h), std::fabs(n_dot_wo * n_dot_h));
    g = (2.0 * g) / std::fabs(wo_dot_h);
    g = rmin(g, 1.0);
    V3 btdf = std::fabs(wi_dot_h * wo_dot_h / (n_dot_wi * n_dot_wo)) *
              (d * (one - f) * g / pow2(eta_t * wi_dot_h + wo_dot_h));
    return cmul(btdf, color);
  }
}

// local_to_world(n) * h : material.rs:316-324 (columns ns, nss, n)
V3 local_to_world_mul(V3 n, V3 h) {
  V3 ns = is_normal(n.x) ? normalize(v3(n.y, -n.x, 0.0)) : normalize(v3(0.0, -n.z, n.y));
  V3 nss = cross(n, ns);
  return {(ns.x * h.x + nss.x * h.y) + n.x * h.z, (ns.y * h.x + nss.y * h.y) + n.y * h.z,
          (ns.z * h.x + nss.z * h.y) + n.z * h.z};
}

bool sample_f(const RptMaterial& m, V3 n, V3 wo, Rng& rng, V3& wi, double& pdf) {
  // material.rs:224-313
  V3 color = from(m.color);
  double m2 = m.roughness * m.roughness;
  double f0 = pow2((m.index - 1.0) / (m.index + 1.0));
  double mean = ((color.x + color.y) + color.z) / 3.0;
  double f = (1.0 - m.metallic) * f0 + m.metallic * mean;
  f = f * (1.0 - 0.2) + 1.0 * 0.2; // glm::mix_scalar(f, 1.0, 0.2)
  double eta_t = dot(wo, n) > 0.0 ? m.index : 1.0 / m.index;

  auto beckmann = [&]() { // :244-254
    double theta = m_atan(std::sqrt(m2 * -m_log(rng.gen_f64())));
    double sin_t, cos_t;
    m_sincos(theta, sin_t, cos_t);
    double x, y;
    rng.unit_circle(x, y);
    V3 h = {x * sin_t, y * sin_t, cos_t};
    return local_to_world_mul(n, h);
  };
  auto beckmann_pdf = [&](V3 h) { // :256-262
    double cos_t = std::fabs(dot(h, n));
    double sin_t = std::sqrt(1.0 - cos_t * cos_t);
    return (1.0 / (PI * m2 * pow3(cos_t))) * m_exp(-pow2(sin_t / cos_t) / m2);
  };

  if (rng.gen_bool(f)) { // :264-267
    V3 h = beckmann();
    V3 refl = wo - h * (dot(h, wo) * 2.0); // glm::reflect_vec(wo, h)
    wi = -refl;
  } else if (!m.transparent) { // :268-273
    double x, y;
    rng.unit_disc(x, y);
    double z = std::sqrt(1.0 - x * x - y * y);
    wi = local_to_world_mul(n, v3(x, y, z));
  } else { // :274-288
    V3 h = beckmann();
    double cos_to = dot(h, wo);
    V3 wo_perp = wo - h * cos_to;
    V3 wi_perp = (-wo_perp) / eta_t;
    double sin2_ti = length2(wi_perp);
    if (sin2_ti > 1.0) return false;
    double cos_ti = std::sqrt(1.0 - sin2_ti);
    wi = -signum(cos_to) * cos_ti * h + wi_perp;
  }

  double p = 0.0;
  {
    V3 h = normalize(wi + wo);
    double p_h = beckmann_pdf(h);
    p += f * p_h / (4.0 * std::fabs(dot(h, wo)));
  }
  if (!m.transparent) {
    p += (1.0 - f) * rmax(dot(wi, n), 0.0) * FRAC_1_PI;
  } else if (sign_positive(dot(wo, n)) != sign_positive(dot(wi, n))) {
    V3 h = normalize(wi * eta_t + wo);
    double p_h = beckmann_pdf(h);
    double h_dot_wo = dot(h, wo);
    double h_dot_wi = dot(h, wi);
    double jacobian = std::fabs(h_dot_wo) / pow2(eta_t * h_dot_wi + h_dot_wo);
    p += (1.0 - f) * p_h * jacobian;
  } else {
    p += 0.0;
  }
  pdf = p;
  return true;
}

// ------------------------------------------------------------------ Light / Environment
struct Light {
  int kind;
  V3 color, vec;
  std::unique_ptr<Shape> shape;
  RptMaterial material;
};

void illuminate(const Light& l, V3 world_pos, Rng& rng, V3& intensity, V3& wi, double& dist) {
  // light.rs:23-47
  switch (l.kind) {
    case RPT_LIGHT_AMBIENT: intensity = l.color; wi = {0, 0, 0}; dist = 0.0; return;
    case RPT_LIGHT_POINT: {
      V3 disp = l.vec - world_pos;
      double len = length(disp);
      intensity = l.color / (len * len);
      wi = disp / len;
      dist = len;
      return;
    }
    case RPT_LIGHT_DIRECTIONAL:
      intensity = l.color;
      wi = -normalize(l.vec);
      dist = INF;
      return;
    default: {
      Sample s = l.shape->sample(world_pos, rng);
      V3 disp = s.v - world_pos;
      double len = length(disp);
      double cosine = rmax(-dot(disp, s.n), 0.0) / len;
      double surface_area = rmax(cosine, 0.0) / (len * len);
      intensity = from(l.material.color) * l.material.emittance * surface_area / s.p;
      wi = disp / len;
      dist = len;
      return;
    }
  }
}

struct Env {
  int kind = RPT_ENV_COLOR;
  V3 color = {0, 0, 0};
  uint32_t width = 0, height = 0;
  std::vector<V3> buf;
  const V3& px(uint64_t i) const {
    static const V3 zero = {0, 0, 0};
    return i < buf.size() ? buf[i] : zero; // the reference would panic out of bounds
  }
  V3 get_color(V3 dir_in) const { // environment.rs:25-52, 72-77
    if (kind == RPT_ENV_COLOR) return color;
    V3 dir = normalize(dir_in);
    double azimuth = m_atan2(dir.z, dir.x) + PI;
    double polar = m_acos(dir.y);
    double x = azimuth / TAU * (double)(width - 1);
    double y = polar / PI * (double)(height - 1);
    auto sat_u32 = [](double v) -> uint32_t { // Rust `as u32` saturates, NaN -> 0
      if (!(v > 0.0)) return 0;
      if (v >= 4294967295.0) return 4294967295u;
      return (uint32_t)v;
    };
    uint32_t x0 = std::min(sat_u32(x), width - 1);
    uint32_t y0 = std::min(sat_u32(y), height - 1);
    double ax = x - (double)x0;
    double ay = y - (double)y0;
    uint64_t w = width;
    V3 top = lerp(px(y0 * w + x0), px(y0 * w + x0 + 1), ax);
    V3 bot = lerp(px((y0 + 1) * w + x0), px((y0 + 1) * w + x0 + 1), ax);
    return lerp(top, bot, ay);
  }
};

// ------------------------------------------------------------------ Scene / Renderer
struct Object {
  std::unique_ptr<Shape> shape;
  RptMaterial material;
};

} // namespace

struct oracle_scene {
  std::vector<Object> objects;
  std::vector<Light> lights;
  Env env;
};

namespace {

Ray cast_ray(const RptCamera& c, double x, double y, Rng& rng) { // camera.rs:64-81
  V3 direction = from(c.direction), up = from(c.up);
  double d = 1.0 / std::tan(c.fov / 2.0);
  V3 right = normalize(cross(direction, up));
  V3 origin = from(c.eye);
  V3 new_dir = d * direction + x * right + y * up;
  if (c.aperture > 0.0) {
    V3 focal_point = origin + normalize(new_dir) * c.focal_distance;
    double a, b;
    rng.unit_disc(a, b);
    origin = origin + (a * right + b * up) * c.aperture;
    new_dir = focal_point - origin;
  }
  return {origin, normalize(new_dir)};
}

struct Tracer {
  const oracle_scene& sc;
  uint32_t max_bounces;
  double* rec = nullptr; // optional per-depth records (8 doubles each)
  int nrec = 0;

  bool closest_hit(const Ray& ray, HitRecord& h, int& obj) const { // renderer.rs:211-220
    obj = -1;
    for (size_t i = 0; i < sc.objects.size(); i++) {
      if (sc.objects[i].shape->intersect(ray, EPSILON, h)) obj = (int)i;
    }
    return obj >= 0;
  }

  V3 sample_lights(const RptMaterial& material, V3 pos, V3 n, V3 wo, Rng& rng) const {
    // renderer.rs:177-204
    V3 color = {0, 0, 0};
    for (const Light& light : sc.lights) {
      if (light.kind == RPT_LIGHT_AMBIENT) {
        color = color + cmul(light.color, from(material.color));
      } else {
        V3 intensity, wi;
        double dist_to_light;
        illuminate(light, pos, rng, intensity, wi, dist_to_light);
        COUNT(shadow_rays);
        HitRecord h;
        int obj;
        tl_shadow = true;
        bool hit = closest_hit({pos, wi}, h, obj);
        tl_shadow = false;
        if (!hit || h.time > dist_to_light) {
          V3 f = bsdf(material, n, wo, wi);
          color = color + cmul(f, intensity) * dot(wi, n);
        }
      }
    }
    return color;
  }

  V3 trace_ray(const Ray& ray, uint32_t num_bounces, Rng& rng) { // renderer.rs:145-174
    COUNT(segments);
    COUNT(closest_rays);
    HitRecord h;
    int obj;
    if (rec) nrec = (int)num_bounces + 1;
    if (!closest_hit(ray, h, obj)) {
      COUNT(misses);
      V3 e = sc.env.get_color(ray.dir);
      if (rec) { double* r = rec + 8 * num_bounces; to(e, r); r[3] = r[4] = r[5] = r[6] = r[7] = 0; }
      return e;
    }
    COUNT(hits);
    V3 world_pos = ray.at(h.time);
    const RptMaterial& material = sc.objects[obj].material;
    V3 wo = -normalize(ray.dir);
    V3 color = material.emittance * from(material.color);
    color = color + sample_lights(material, world_pos, h.normal, wo, rng);
    if (rec) { double* r = rec + 8 * num_bounces; to(color, r); r[3] = r[4] = r[5] = r[6] = r[7] = 0; }
    if (num_bounces < max_bounces) {
      V3 wi;
      double pdf;
      if (sample_f(material, h.normal, wo, rng, wi, pdf)) {
        V3 f = bsdf(material, h.normal, wo, wi);
        double abscos = std::fabs(dot(wi, h.normal));
        if (rec) { double* r = rec + 8 * num_bounces; to(f, r + 3); r[6] = 1.0 / pdf; r[7] = abscos; }
        V3 next = trace_ray({world_pos, wi}, num_bounces + 1, rng);
        V3 indirect = 1.0 / pdf * cmul(f, next) * abscos;
        color.x += rmin(indirect.x, FIREFLY_CLAMP);
        color.y += rmin(indirect.y, FIREFLY_CLAMP);
        color.z += rmin(indirect.z, FIREFLY_CLAMP);
      }
    }
    return color;
  }
};

inline void pixel_ndc(const RptRenderParams& p, uint32_t x, uint32_t y, double& xn, double& yn,
                      double& dim) { // renderer.rs:132-134
  dim = (double)std::max(p.width, p.height);
  xn = ((double)(2 * x + 1) - (double)p.width) / dim;
  yn = ((double)(2 * (p.height - y) - 1) - (double)p.height) / dim;
}

V3 sample_once(Tracer& tr, const RptCamera& cam, const RptRenderParams& p, uint32_t x,
               uint32_t y, uint64_t sample) { // body of the loop renderer.rs:136-140
  double xn, yn, dim;
  pixel_ndc(p, x, y, xn, yn, dim);
  Rng rng(p.seed, y * p.width + x, sample);
  double dx = rng.gen_range(-1.0 / dim, 1.0 / dim);
  double dy = rng.gen_range(-1.0 / dim, 1.0 / dim);
  COUNT(samples);
  return tr.trace_ray(cast_ray(cam, xn + dx, yn + dy, rng), 0, rng);
}

V3 get_color(Tracer& tr, const RptCamera& cam, const RptRenderParams& p, uint32_t x, uint32_t y) {
  // renderer.rs:131-142
  V3 color = {0, 0, 0};
  for (uint32_t i = 0; i < p.iterations; i++)
    color = color + sample_once(tr, cam, p, x, y, p.sample_index_base + i);
  return color / (double)p.iterations * std::pow(2.0, p.exposure_value);
}

inline bool pixel_in_part(const RptRenderParams& p, uint32_t x, uint32_t y) {
  if (p.part_count <= 1) return true;
  uint32_t tw = p.tile_width ? p.tile_width : 32, th = p.tile_height ? p.tile_height : 8;
  uint32_t tiles_x = (p.width + tw - 1) / tw;
  uint32_t tile = (y / th) * tiles_x + (x / tw);
  return tile % p.part_count == p.part_index;
}

void flatten_kd(const KdNode& n, uint32_t id, uint32_t depth, std::vector<double>& split,
                std::vector<uint32_t>& info, std::vector<uint32_t>& a, std::vector<uint32_t>& b,
                std::vector<uint32_t>& refs, uint32_t& max_depth) {
  max_depth = std::max(max_depth, depth);
  if (n.axis == 3) {
    info[id] = 3;
    split[id] = 0.0;
    a[id] = (uint32_t)refs.size();
    b[id] = (uint32_t)n.indices.size();
    for (size_t i : n.indices) refs.push_back((uint32_t)i);
    return;
  }
  uint32_t l = (uint32_t)split.size();
  split.push_back(0); split.push_back(0);
  info.push_back(0); info.push_back(0);
  a.push_back(0); a.push_back(0);
  b.push_back(0); b.push_back(0);
  info[id] = (uint32_t)n.axis;
  split[id] = n.value;
  a[id] = l;
  b[id] = 0;
  flatten_kd(*n.left, l, depth + 1, split, info, a, b, refs, max_depth);
  flatten_kd(*n.right, l + 1, depth + 1, split, info, a, b, refs, max_depth);
}

template <class T> T* dup(const std::vector<T>& v) {
  T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
  std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

} // namespace

// =================================================================== C entry points
extern "C" {

int oracle_scene_create(const RptScene* d, oracle_scene** out) {
  if (!d || !out) return RPTGPU_E_INVALID_ARGUMENT;
  auto s = std::make_unique<oracle_scene>();
  for (uint64_t i = 0; i < d->num_objects; i++) {
    Object o;
    o.shape = make_shape(d->objects[i].shape);
    if (!o.shape) return RPTGPU_E_UNSUPPORTED_SHAPE;
    o.material = d->objects[i].material;
    s->objects.push_back(std::move(o));
  }
  for (uint64_t i = 0; i < d->num_lights; i++) {
    const RptLight& l = d->lights[i];
    Light L;
    L.kind = l.kind;
    L.color = from(l.color);
    L.vec = from(l.vec);
    L.material = l.object.material;
    if (l.kind == RPT_LIGHT_OBJECT) {
      L.shape = make_shape(l.object.shape);
      if (!L.shape) return RPTGPU_E_UNSUPPORTED_SHAPE;
    }
    s->lights.push_back(std::move(L));
  }
  s->env.kind = d->environment.kind;
  s->env.color = from(d->environment.color);
  if (d->environment.kind == RPT_ENV_HDRI) {
    s->env.width = d->environment.width;
    s->env.height = d->environment.height;
    uint64_t n = (uint64_t)s->env.width * s->env.height;
    s->env.buf.resize(n);
    for (uint64_t i = 0; i < n; i++) s->env.buf[i] = from(d->environment.texels + 3 * i);
  }
  *out = s.release();
  return RPTGPU_OK;
}

void oracle_scene_destroy(oracle_scene* s) { delete s; }

int oracle_render(const oracle_scene* s, const RptCamera* cam, const RptRenderParams* p,
                  int threads, double* out_rgb, OracleCounters* counters) {
  if (!s || !cam || !p || !out_rgb) return RPTGPU_E_INVALID_ARGUMENT;
  if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  std::atomic<uint32_t> next_row{0};
  std::vector<Counters> cnts(threads);
  auto work = [&](int tid) {
    tl_cnt = counters ? &cnts[tid] : nullptr;
    Tracer tr{*s, p->max_bounces};
    for (;;) { // one task per image row, claimed dynamically (renderer.rs:118-127)
      uint32_t y = next_row.fetch_add(1);
      if (y >= p->height) break;
      for (uint32_t x = 0; x < p->width; x++) {
        V3 c = {0, 0, 0};
        if (pixel_in_part(*p, x, y)) c = get_color(tr, *cam, *p, x, y);
        to(c, out_rgb + 3 * ((uint64_t)y * p->width + x));
      }
    }
    tl_cnt = nullptr;
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();
  if (counters) {
    Counters total;
    for (auto& c : cnts) total.add(c);
    *counters = total;
  }
  return RPTGPU_OK;
}

int oracle_trace_sample(const oracle_scene* s, const RptCamera* cam, const RptRenderParams* p,
                        uint32_t x, uint32_t y, uint64_t sample, double* out_rgb, double* rec,
                        int* out_nrec) {
  if (!s || !cam || !p || !out_rgb) return RPTGPU_E_INVALID_ARGUMENT;
  Tracer tr{*s, p->max_bounces};
  tr.rec = rec;
  V3 c = sample_once(tr, *cam, *p, x, y, sample);
  to(c, out_rgb);
  if (out_nrec) *out_nrec = tr.nrec;
  return RPTGPU_OK;
}

int oracle_closest_hit(const oracle_scene* s, uint64_t n, const double* origins,
                       const double* dirs, double* out_t, double* out_normal,
                       int32_t* out_object, OracleCounters* counters) {
  if (!s) return RPTGPU_E_INVALID_ARGUMENT;
  Counters c;
  tl_cnt = counters ? &c : nullptr;
  Tracer tr{*s, 0};
  for (uint64_t i = 0; i < n; i++) {
    HitRecord h;
    int obj;
    COUNT(closest_rays);
    tr.closest_hit({from(origins + 3 * i), from(dirs + 3 * i)}, h, obj);
    out_t[i] = h.time;
    to(h.normal, out_normal + 3 * i);
    out_object[i] = obj;
  }
  tl_cnt = nullptr;
  if (counters) *counters = c;
  return RPTGPU_OK;
}

int oracle_camera_ray(const RptCamera* cam, const RptRenderParams* p, uint32_t x, uint32_t y,
                      uint64_t sample, double* out6) {
  double xn, yn, dim;
  pixel_ndc(*p, x, y, xn, yn, dim);
  Rng rng(p->seed, y * p->width + x, sample);
  double dx = rng.gen_range(-1.0 / dim, 1.0 / dim);
  double dy = rng.gen_range(-1.0 / dim, 1.0 / dim);
  Ray r = cast_ray(*cam, xn + dx, yn + dy, rng);
  to(r.origin, out6);
  to(r.dir, out6 + 3);
  return RPTGPU_OK;
}

void oracle_bsdf(const RptMaterial* m, const double* n, const double* wo, const double* wi,
                 double* out3) {
  to(bsdf(*m, from(n), from(wo), from(wi)), out3);
}

int oracle_sample_f(const RptMaterial* m, const double* n, const double* wo, uint64_t seed,
                    uint32_t pixel, uint64_t sample, uint32_t* draw, double* out_wi,
                    double* out_pdf) {
  Rng rng(seed, pixel, sample);
  rng.draw = *draw;
  V3 wi = {0, 0, 0};
  double pdf = 0;
  bool some = sample_f(*m, from(n), from(wo), rng, wi, pdf);
  *draw = rng.draw;
  to(wi, out_wi);
  *out_pdf = pdf;
  return some ? 1 : 0;
}

int oracle_illuminate(const RptLight* light, const double* pos, uint64_t seed, uint32_t pixel,
                      uint64_t sample, uint32_t* draw, double* out7) {
  Light L;
  L.kind = light->kind;
  L.color = from(light->color);
  L.vec = from(light->vec);
  L.material = light->object.material;
  if (light->kind == RPT_LIGHT_OBJECT) {
    L.shape = make_shape(light->object.shape);
    if (!L.shape) return RPTGPU_E_UNSUPPORTED_SHAPE;
    if (!L.shape->bounded()) return RPTGPU_E_UNIMPLEMENTED_SAMPLE;
  }
  Rng rng(seed, pixel, sample);
  rng.draw = *draw;
  V3 intensity, wi;
  double dist;
  illuminate(L, from(pos), rng, intensity, wi, dist);
  *draw = rng.draw;
  to(intensity, out7);
  to(wi, out7 + 3);
  out7[6] = dist;
  return RPTGPU_OK;
}

void oracle_env_color(const RptEnvironment* e, const double* dir, double* out3) {
  Env env;
  env.kind = e->kind;
  env.color = from(e->color);
  if (e->kind == RPT_ENV_HDRI) {
    env.width = e->width;
    env.height = e->height;
    uint64_t n = (uint64_t)e->width * e->height;
    env.buf.resize(n);
    for (uint64_t i = 0; i < n; i++) env.buf[i] = from(e->texels + 3 * i);
  }
  to(env.get_color(from(dir)), out3);
}

int oracle_shape_intersect(const RptShape* shape, const double* origin, const double* dir,
                           double t_min, double* inout_time, double* out_normal) {
  auto s = make_shape(*shape);
  if (!s) return RPTGPU_E_UNSUPPORTED_SHAPE;
  HitRecord h;
  h.time = *inout_time;
  bool r = s->intersect({from(origin), from(dir)}, t_min, h);
  *inout_time = h.time;
  to(h.normal, out_normal);
  return r ? 1 : 0;
}

int oracle_shape_sample(const RptShape* shape, const double* target, uint64_t seed,
                        uint32_t pixel, uint64_t sample, uint32_t* draw, double* out7) {
  auto s = make_shape(*shape);
  if (!s) return RPTGPU_E_UNSUPPORTED_SHAPE;
  if (!s->bounded()) return RPTGPU_E_UNIMPLEMENTED_SAMPLE;
  Rng rng(seed, pixel, sample);
  rng.draw = *draw;
  Sample r = s->sample(from(target), rng);
  *draw = rng.draw;
  to(r.v, out7);
  to(r.n, out7 + 3);
  out7[6] = r.p;
  return RPTGPU_OK;
}

void oracle_monomial_closest_point(double height, const double* point, int steps, double* out3) {
  Monomial m;
  m.height = height;
  to(m.closest_point(from(point), steps), out3);
}

void oracle_bbox_intersect(const double* box6, const double* origin, const double* dir,
                           double* out_min, double* out_max) {
  BBox b{from(box6), from(box6 + 3)};
  b.intersect({from(origin), from(dir)}, *out_min, *out_max);
}

int oracle_kdtree_build(const double* boxes, uint64_t n, RptKdTree* out) {
  if (!out || (!boxes && n)) return RPTGPU_E_INVALID_ARGUMENT;
  std::vector<BBox> all(n);
  std::vector<size_t> idx(n);
  for (uint64_t i = 0; i < n; i++) {
    all[i] = {from(boxes + 6 * i), from(boxes + 6 * i + 3)};
    idx[i] = i;
  }
  auto root = construct(all, std::move(idx));
  std::vector<double> split(1);
  std::vector<uint32_t> info(1), a(1), b(1), refs;
  uint32_t max_depth = 0;
  flatten_kd(*root, 0, 0, split, info, a, b, refs, max_depth);
  out->num_nodes = split.size();
  out->num_refs = refs.size();
  out->max_depth = max_depth;
  out->regular = 0; // not computed by the restatement
  out->split = dup(split);
  out->info = dup(info);
  out->a = dup(a);
  out->b = dup(b);
  out->refs = dup(refs);
  return RPTGPU_OK;
}

void oracle_kdtree_free(RptKdTree* t) {
  if (!t) return;
  std::free(t->split); std::free(t->info); std::free(t->a); std::free(t->b); std::free(t->refs);
  std::memset(t, 0, sizeof(*t));
}

void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32_10(ctr, key, out);
}

uint64_t oracle_rng_u64(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t draw) {
  Rng rng(seed, pixel, sample);
  rng.draw = draw;
  return rng.next_u64();
}

int oracle_rng_sample(int kind, double lo, double hi, uint64_t seed, uint32_t pixel,
                      uint64_t sample, uint32_t* draw, double* out2) {
  Rng rng(seed, pixel, sample);
  rng.draw = *draw;
  out2[0] = out2[1] = 0;
  switch (kind) {
    case 0: out2[0] = rng.gen_f64(); break;
    case 1: out2[0] = rng.gen_range(lo, hi); break;
    case 2: out2[0] = rng.gen_bool(lo) ? 1.0 : 0.0; break;
    case 3: out2[0] = (double)rng.gen_index((uint64_t)lo); break;
    case 4: rng.unit_disc(out2[0], out2[1]); break;
    case 5: rng.unit_circle(out2[0], out2[1]); break;
    default: return RPTGPU_E_INVALID_ARGUMENT;
  }
  *draw = rng.draw;
  return RPTGPU_OK;
}

void oracle_math_eval(int fn, uint64_t n, const double* x, const double* y, double* out) {
  for (uint64_t i = 0; i < n; i++) {
    switch (fn) {
      case 0: out[i] = rpt_exp(x[i]); break;
      case 1: out[i] = rpt_log(x[i]); break;
      case 2: out[i] = rpt_atan(x[i]); break;
      case 3: { double s, c; rpt_sincos_pio2(x[i], &s, &c); out[i] = s; break; }
      case 4: { double s, c; rpt_sincos_pio2(x[i], &s, &c); out[i] = c; break; }
      case 5: out[i] = rpt_acos(x[i]); break;
      default: out[i] = rpt_atan2(y[i], x[i]); break;
    }
  }
}

void oracle_hex_color(uint32_t x, double* out3) { // color.rs:10-15
  const double g = 2.2;
  double r = (double)((x >> 16) & 0xff) / 255.0;
  double gg = (double)((x >> 8) & 0xff) / 255.0;
  double b = (double)(x & 0xff) / 255.0;
  out3[0] = std::pow(r, g);
  out3[1] = std::pow(gg, g);
  out3[2] = std::pow(b, g);
}

void oracle_color_bytes(const double* c, uint8_t* out3) { // color.rs:18-24
  for (int i = 0; i < 3; i++) {
    double v = std::pow(rmin(rmax(c[i], 0.0), 1.0), 1.0 / 2.2) * 255.0;
    // Rust `as u8`: saturating, truncating, NaN -> 0
    out3[i] = !(v > 0.0) ? 0 : (v >= 255.0 ? 255 : (uint8_t)v);
  }
}

void oracle_buffer_image(uint32_t w, uint32_t h, uint32_t radius, uint32_t nb,
                         const double* const* batches, uint8_t* out) {
  // buffer.rs:43-56 with get_filtered_color buffer.rs:75-93
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) {
      V3 color = {0, 0, 0};
      uint64_t count = 0;
      uint32_t i0 = x >= radius ? x - radius : 0, j0 = y >= radius ? y - radius : 0;
      for (uint32_t i = i0; i <= x + radius; i++)
        for (uint32_t j = j0; j <= y + radius; j++)
          if (i < w && j < h) {
            uint64_t index = (uint64_t)j * w + i;
            V3 sum = {0, 0, 0}; // iter().sum::<Color>() starts from zero
            for (uint32_t b = 0; b < nb; b++) sum = sum + from(batches[b] + 3 * index);
            color = color + sum;
            count += nb;
          }
      V3 c = color / (double)count;
      double cc[3] = {c.x, c.y, c.z};
      oracle_color_bytes(cc, out + 3 * ((uint64_t)y * w + x));
    }
}

double oracle_buffer_variance(uint32_t w, uint32_t h, uint32_t nb, const double* const* batches) {
  // buffer.rs:59-73
  double variance = 0.0, count = 0.0;
  for (uint64_t p = 0; p < (uint64_t)w * h; p++) {
    V3 sum = {0, 0, 0};
    for (uint32_t b = 0; b < nb; b++) sum = sum + from(batches[b] + 3 * p);
    V3 mean = sum / (double)nb;
    double sum_of_squares = 0.0;
    for (uint32_t b = 0; b < nb; b++) sum_of_squares += length2(from(batches[b] + 3 * p) - mean);
    variance += sum_of_squares / ((double)nb - 1.0);
    count += 1.0;
  }
  return variance / count;
}

} // extern "C"
