// host_scene.cpp — scene hand-off: kd-tree construction and flattening (host, untimed).
//
// Reference behaviour reproduced here (the device then only READS the result):
//   KdTree::new / construct / median          src/kdtree.rs:108-119, 235-355
//   Triangle::bounding_box                    src/shape/mesh.rs:40-45
//   Sphere / Cube bounding boxes              src/shape/sphere.rs:66-73, src/shape/cube.rs:10-17
//   Transformed::bounding_box (8 corners)     src/shape.rs:153-176
// Compiled with -ffp-contract=off so the split planes are the same doubles the reference
// computes (medians are sums of two doubles halved).
#include "host_scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <future>
#include <functional>
#include <map>
#include <thread>
#include <cstdlib>
#include <cstdio>
#include <system_error>
#include <sched.h>

namespace rpthost {

namespace {

// CPUs this process may actually run on: the affinity mask and the cgroup v2 quota, not the machine's core count
// (a GPU box reports 256 logical CPUs and grants 16)
int usable_cpus() {
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    double period = 0.0;
    if (std::fscanf(f, "%31s %lf", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0.0)
      n = std::min(n, std::max(1, (int)(std::atof(quota) / period)));
    std::fclose(f);
  }
  // one process per GPU: the ranks of a node build their scenes at the same time on the same cores.  LOCAL_WORLD_SIZE is
  // what torch.distributed.run exports; RPTGPU_LOCAL_RANKS says the same for other launchers.
  for (const char* name : {"RPTGPU_LOCAL_RANKS", "LOCAL_WORLD_SIZE"})
    if (const char* e = std::getenv(name)) {
      const int ranks = std::atoi(e);
      if (ranks > 1) n = std::max(1, n / ranks);
      break;
    }
  return n;
}

// fn(begin, end) over [0, n) in contiguous pieces on the usable CPUs; small ranges, or a box that cannot start threads,
// run on the caller's.  For the per-triangle / per-leaf-entry loops of the flattening (independent items).
// Host threads of the flattening and the kd build: RptSceneOptions::build_threads while a scene is being made
// (flatten_scene sets it for its thread; the environment's override is already folded in, api_scene.cpp), else — the
// handle-less rptgpu_kdtree_build — RPTGPU_BUILD_THREADS; 0 = the usable cores.
thread_local int t_build_threads = 0;
int forced_build_threads() {
  if (t_build_threads > 0) return t_build_threads;
  if (const char* e = std::getenv("RPTGPU_BUILD_THREADS")) return std::max(1, std::atoi(e));
  return 0;
}
template <class F> void parallel_for(size_t n, size_t min_per_thread, F fn) {
  int threads = (int)std::min<size_t>((size_t)std::min(usable_cpus(), 32), n / std::max<size_t>(min_per_thread, 1));
  if (const int f = forced_build_threads()) threads = std::min(threads, f);
  if (threads <= 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> pool;
  const size_t step = (n + (size_t)threads - 1) / (size_t)threads;
  size_t done = 0;
  try {
    for (; done < n; done += step) {
      const size_t b = done, e = std::min(n, done + step);
      pool.emplace_back([=] { fn(b, e); });
    }
  } catch (const std::system_error&) { // thread limit: the rest here
    fn(done, n);
  }
  for (std::thread& t : pool) t.join();
}

constexpr double SCORE_THRESHOLD = 0.85; // kdtree.rs:6

// median of the multiset `v` as `median(&sorted)` (kdtree.rs:347-355) would return it, without
// the full sort the reference does: only the two middle order statistics matter.
// The reference's sort is STABLE and compares -0.0 == +0.0 (kdtree.rs:251-254): among equal entries the pushed order
// (per index: p_min, p_max) survives.  Only one thing can see it — the SIGN of a zero median when both middle entries
// are zeros: (-0 + -0) / 2 = -0, any other pair of zeros +0.  `both_zero` reports that case; median_zero_sign then
// finds WHICH two pushed entries the stable sort puts in the middle.
double median_of(std::vector<double>& v, bool& both_zero) {
  size_t n = v.size();
  size_t mid = n / 2;
  std::nth_element(v.begin(), v.begin() + mid, v.end());
  double hi = v[mid];
  both_zero = false;
  if (n % 2 == 1) return hi;
  double lo = *std::max_element(v.begin(), v.begin() + mid);
  both_zero = hi == 0.0 && lo == 0.0;
  return (hi + lo) / 2.0;
}
// the median (a zero) of the 2n pushed values `get(0), get(1), ...`, whose stable sort has zeros at both middle places
template <class Get> double median_zero_sign(size_t n2, Get get) {
  const size_t mid = n2 / 2;
  size_t below = 0; // entries that sort before every zero
  for (size_t j = 0; j < n2; j++) below += get(j) < 0.0 ? 1 : 0;
  const size_t want0 = mid - 1 - below, want1 = mid - below; // ranks of the two middle entries among the zeros, pushed order
  size_t z = 0;
  bool neg0 = false, neg1 = false;
  for (size_t j = 0; j < n2 && z <= want1; j++) {
    const double v = get(j);
    if (v == 0.0) {
      if (z == want0) neg0 = std::signbit(v);
      if (z == want1) neg1 = std::signbit(v);
      z++;
    }
  }
  return neg0 && neg1 ? -0.0 : 0.0;
}

struct Builder {
  const std::vector<Box>& boxes;
  KdBuild& out;
  std::vector<double> xs, ys, zs; // scratch

  void make_leaf(uint32_t id, const std::vector<uint32_t>& idx) {
    out.nodes[id].split = 0.0;
    out.nodes[id].a = (uint32_t)out.refs.size();
    out.nodes[id].ib = 3u | ((uint32_t)idx.size() << 2);
    out.refs.insert(out.refs.end(), idx.begin(), idx.end());
  }

  // cell = the node's box as the reference derives it while traversing: the root bounds cut by the
  // ancestors' split planes (BoundingBox::split, kdtree.rs:71-86)
  // the decision of construct (kdtree.rs:235-319) for the primitives `idx`: leaf, or (axis, value) and the two
  // index lists (straddlers in both, kdtree.rs:270-281)
  void decide(const std::vector<uint32_t>& idx, bool& leaf, int& split_dir, double& value, std::vector<uint32_t>& left,
              std::vector<uint32_t>& right) {
    size_t n = idx.size();
    leaf = true;
    if (n < 16) return; // kdtree.rs:236-238
    xs.clear(); ys.clear(); zs.clear();
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i : idx) {
      const Box& b = boxes[i];
      xs.push_back(b.lo[0]); xs.push_back(b.hi[0]);
      ys.push_back(b.lo[1]); ys.push_back(b.hi[1]);
      zs.push_back(b.lo[2]); zs.push_back(b.hi[2]);
      for (int k = 0; k < 3; k++) {
        lo[k] = std::fmin(lo[k], b.lo[k]);
        hi[k] = std::fmax(hi[k], b.hi[k]);
      }
    }
    bool bz[3];
    double m[3] = {median_of(xs, bz[0]), median_of(ys, bz[1]), median_of(zs, bz[2])}; // kdtree.rs:252-255
    for (int k = 0; k < 3; k++)
      if (bz[k]) m[k] = median_zero_sign(2 * n, [&](size_t j) { const Box& b = boxes[idx[j >> 1]]; return (j & 1) ? b.hi[k] : b.lo[k]; });
    size_t s[3];
    for (int dim = 0; dim < 3; dim++) { // partition_score kdtree.rs:257-268
      size_t l = 0, r = 0;
      for (uint32_t i : idx) {
        if (boxes[i].lo[dim] <= m[dim]) l++;
        if (boxes[i].hi[dim] >= m[dim]) r++;
      }
      s[dim] = std::max(l, r);
    }
    size_t threshold = (size_t)((double)n * SCORE_THRESHOLD); // kdtree.rs:286
    if (std::min(std::min(s[0], s[1]), s[2]) >= threshold) return;
    split_dir = -1; // kdtree.rs:291-319
    double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    if (ex > ey && ex > ez) {
      if (s[0] < threshold) split_dir = 0;
    } else if (ey > ez) {
      if (s[1] < threshold) split_dir = 1;
    } else if (s[2] < threshold) {
      split_dir = 2;
    }
    if (split_dir == -1) {
      if (s[0] < s[1] && s[0] < s[2]) split_dir = 0;
      else if (s[1] < s[2]) split_dir = 1;
      else split_dir = 2;
    }
    value = m[split_dir];
    left.clear(); right.clear();
    for (uint32_t i : idx) { // partition kdtree.rs:270-281 (straddlers go to both)
      if (boxes[i].lo[split_dir] <= value) left.push_back(i);
      if (boxes[i].hi[split_dir] >= value) right.push_back(i);
    }
    leaf = false;
  }

  // cell = the node's box as the reference derives it while traversing: the root bounds cut by the
  // ancestors' split planes (BoundingBox::split, kdtree.rs:71-86)
  void construct(uint32_t id, std::vector<uint32_t>& idx, uint32_t depth, Box cell) {
    out.max_depth = std::max(out.max_depth, depth);
    bool leaf;
    int split_dir = -1;
    double value = 0.0;
    std::vector<uint32_t> left, right;
    decide(idx, leaf, split_dir, value, left, right);
    if (leaf) {
      make_leaf(id, idx);
      return;
    }
    std::vector<uint32_t>().swap(idx); // release before recursing
    uint32_t l = (uint32_t)out.nodes.size();
    out.nodes.push_back({});
    out.nodes.push_back({});
    out.nodes[id].split = value;
    out.nodes[id].a = l;
    out.nodes[id].ib = (uint32_t)split_dir;
    // the median of the primitives' box edges can fall outside the cell (straddlers reach beyond
    // it); the device's compact traversal assumes it does not and is disabled for such trees
    if (!(value >= cell.lo[split_dir] && value <= cell.hi[split_dir])) out.regular = false;
    Box cl = cell, cr = cell;
    cl.hi[split_dir] = value;
    cr.lo[split_dir] = value;
    construct(l, left, depth + 1, cl);
    construct(l + 1, right, depth + 1, cr);
  }
};

// ---- parallel construction -----------------------------------------------------------------------------------
// construct() numbers nodes in the order a sequential depth-first build creates them: a node's two children are
// allocated as a pair when the node is split, then the whole left subtree, then the whole right subtree; leaf
// entries are appended to refs[] in the same order.  So a subtree built on its own (root = local node 0) can be
// spliced in afterwards by shifting indices: the numbering — hence the tree the device traverses and every
// fixture — is identical whatever the number of threads.  Subtrees above PAR_MIN_PRIMS primitives in the first
// PAR_MAX_DEPTH levels fork their left child onto another thread (std::async); everything below runs the
// sequential builder.  (The reference builds sequentially with three full sorts per node, kdtree.rs:252-255.)
constexpr size_t PAR_MIN_PRIMS = 8192;
constexpr uint32_t PAR_MAX_DEPTH = 5;

void splice(KdBuild& dst, uint32_t dst_root, const KdBuild& sub) {
  // sub's node 0 becomes dst.nodes[dst_root]; sub's nodes 1.. are appended; child and ref indices shift
  const uint32_t node_shift = (uint32_t)dst.nodes.size() - 1u, ref_shift = (uint32_t)dst.refs.size();
  auto fix = [&](rptdev::KdNode n) {
    if ((n.ib & 3u) == 3u) n.a += ref_shift;
    else n.a += node_shift;
    return n;
  };
  dst.nodes[dst_root] = fix(sub.nodes[0]);
  for (size_t i = 1; i < sub.nodes.size(); i++) dst.nodes.push_back(fix(sub.nodes[i]));
  dst.refs.insert(dst.refs.end(), sub.refs.begin(), sub.refs.end());
  dst.max_depth = std::max(dst.max_depth, sub.max_depth);
  dst.regular = dst.regular && sub.regular;
}

void build_subtree(const std::vector<Box>& boxes, std::vector<uint32_t>& idx, uint32_t depth, Box cell, KdBuild& out, int threads);

// one split at the top of a large subtree, children built concurrently, then spliced in sequential order
void build_forked(const std::vector<Box>& boxes, std::vector<uint32_t>& idx, uint32_t depth, Box cell, KdBuild& out, int threads) {
  // the split decision is the sequential builder's own (Builder::decide)
  struct { std::vector<uint32_t> left, right; int dir = -1; double value = 0.0; bool leaf = false; } pb;
  {
    KdBuild scratch;
    Builder probe{boxes, scratch, {}, {}, {}};
    probe.decide(idx, pb.leaf, pb.dir, pb.value, pb.left, pb.right);
  }
  out.nodes.clear(); out.refs.clear(); out.max_depth = depth; out.regular = true;
  out.nodes.push_back({});
  if (pb.leaf) {
    out.nodes[0].split = 0.0; out.nodes[0].a = 0; out.nodes[0].ib = 3u | ((uint32_t)idx.size() << 2);
    out.refs = idx;
    return;
  }
  std::vector<uint32_t>().swap(idx);
  out.nodes.push_back({}); out.nodes.push_back({});
  out.nodes[0].split = pb.value; out.nodes[0].a = 1; out.nodes[0].ib = (uint32_t)pb.dir;
  if (!(pb.value >= cell.lo[pb.dir] && pb.value <= cell.hi[pb.dir])) out.regular = false;
  Box cl = cell, cr = cell;
  cl.hi[pb.dir] = pb.value; cr.lo[pb.dir] = pb.value;
  KdBuild lb, rb;
  std::future<void> fut;
  bool forked = false;
  try { // thread creation can fail under container pid / thread limits: build both children here then (same tree)
    fut = std::async(std::launch::async, [&] { build_subtree(boxes, pb.left, depth + 1, cl, lb, threads / 2); });
    forked = true;
  } catch (const std::system_error&) {
  }
  if (forked) {
    build_subtree(boxes, pb.right, depth + 1, cr, rb, threads - threads / 2);
    fut.get();
  } else {
    build_subtree(boxes, pb.left, depth + 1, cl, lb, 1);
    build_subtree(boxes, pb.right, depth + 1, cr, rb, 1);
  }
  splice(out, 1, lb);
  splice(out, 2, rb);
}

void build_subtree(const std::vector<Box>& boxes, std::vector<uint32_t>& idx, uint32_t depth, Box cell, KdBuild& out, int threads) {
  if (threads > 1 && idx.size() >= PAR_MIN_PRIMS && depth < PAR_MAX_DEPTH) {
    build_forked(boxes, idx, depth, cell, out, threads);
    return;
  }
  out.nodes.clear(); out.refs.clear(); out.max_depth = 0; out.regular = true;
  out.nodes.push_back({});
  Builder b{boxes, out, {}, {}, {}};
  b.construct(0, idx, depth, cell);
}

} // namespace

void kd_build(const std::vector<Box>& boxes, KdBuild& out, int threads) {
  std::vector<uint32_t> idx(boxes.size());
  Box root;
  for (int k = 0; k < 3; k++) { root.lo[k] = INFINITY; root.hi[k] = -INFINITY; }
  for (size_t i = 0; i < boxes.size(); i++) {
    idx[i] = (uint32_t)i;
    for (int k = 0; k < 3; k++) {
      root.lo[k] = std::fmin(root.lo[k], boxes[i].lo[k]);
      root.hi[k] = std::fmax(root.hi[k], boxes[i].hi[k]);
    }
  }
  if (threads <= 0) {
    threads = usable_cpus();
    if (const int f = forced_build_threads()) threads = f;
    threads = std::min(threads, 32);
  }
  build_subtree(boxes, idx, 0, root, out, threads);
}

// ------------------------------------------------------------------------- flattening
namespace {

// column-major 4x4 * (v,1), accumulated column by column (nalgebra gemv order)
void xf_point(const double* m, const double* v, double* r) {
  for (int k = 0; k < 3; k++) r[k] = ((m[k] * v[0] + m[4 + k] * v[1]) + m[8 + k] * v[2]) + m[12 + k] * 1.0;
}

Box merge(const Box& a, const Box& b) { // BoundingBox::merge kdtree.rs:46-51
  Box r;
  for (int k = 0; k < 3; k++) {
    r.lo[k] = std::fmin(a.lo[k], b.lo[k]);
    r.hi[k] = std::fmax(a.hi[k], b.hi[k]);
  }
  return r;
}

Box empty_box() { // BoundingBox::default kdtree.rs:35-42
  Box b;
  for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
  return b;
}

Box transformed_box(const Box& b, const double* m) { // shape.rs:153-176
  Box r = empty_box();
  for (int ix = 0; ix < 2; ix++)
    for (int iy = 0; iy < 2; iy++)
      for (int iz = 0; iz < 2; iz++) {
        double v[3] = {ix ? b.hi[0] : b.lo[0], iy ? b.hi[1] : b.lo[1], iz ? b.hi[2] : b.lo[2]};
        double c[3];
        xf_point(m, v, c);
        Box p;
        for (int k = 0; k < 3; k++) p.lo[k] = p.hi[k] = c[k];
        r = merge(r, p);
      }
  return r;
}

// mesh.rs:50-51 and :64-69, same expression order as the reference (nalgebra dot = (a+b)+c,
// normalize = component / norm); this file is compiled with -ffp-contract=off
void fill_trix(const RptTriangle& t, rptdev::TriX& x) {
  double d0[3], d1[3], c[3];
  for (int k = 0; k < 3; k++) { d0[k] = t.v2[k] - t.v1[k]; d1[k] = t.v3[k] - t.v1[k]; }
  c[0] = d0[1] * d1[2] - d0[2] * d1[1];
  c[1] = d0[2] * d1[0] - d0[0] * d1[2];
  c[2] = d0[0] * d1[1] - d0[1] * d1[0];
  double len = std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
  for (int k = 0; k < 3; k++) { x.pn[k] = c[k] / len; x.v1[k] = t.v1[k]; x.d0[k] = d0[k]; x.d1[k] = d1[k]; }
  x.d00 = (d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2];
  x.d01 = (d0[0] * d1[0] + d0[1] * d1[1]) + d0[2] * d1[2];
  x.d11 = (d1[0] * d1[0] + d1[1] * d1[1]) + d1[2] * d1[2];
  x.denom = x.d00 * x.d11 - x.d01 * x.d01;
}

// Conservative 16-bit boxes of a mesh's leaf entries (device_types.h LeafBox).  Grid: 65529 steps across the
// tree's bounds per axis plus two steps of padding on either side; a minimum is rounded down and a maximum up, then
// both move one more step outwards, so the decoded box contains the triangle's true box with a margin of at least
// (1 - 1e-12) steps on every side —
// orders of magnitude more than the rounding of the decode and of the slab arithmetic on the device.  A triangle
// whose barycentric system is ill-conditioned (sliver: rounding in mesh.rs:64-73 could accept a point that is not
// near the triangle) or that has a non-finite vertex gets the whole grid, i.e. it is never filtered.  GROUP trees get
// the boxes of their children the same way (a child's hit point lies on the child, hence in its bounding box).
rptdev::LeafBox quantise_box(const Box& b, const double* qlo, const double* qscale, bool full) {
  uint32_t q[6];
  for (int k = 0; k < 3 && !full; k++) {
    double a = std::floor((b.lo[k] - qlo[k]) / qscale[k]) - 1.0;
    double c = std::ceil((b.hi[k] - qlo[k]) / qscale[k]) + 1.0;
    if (!(a == a) || !(c == c)) { full = true; break; }
    q[k] = (uint32_t)std::fmin(std::fmax(a, 0.0), 65535.0);
    q[3 + k] = (uint32_t)std::fmin(std::fmax(c, 0.0), 65535.0);
  }
  if (full) { q[0] = q[1] = q[2] = 0; q[3] = q[4] = q[5] = 65535; }
  // stored per axis as centre and half-extent (device_types.h): c = floor of the middle, h = hi - c >= c - lo, so
  // [c - h, c + h] contains [lo, hi] and is at most one step wider on the low side (c - h may be -1: the decode is
  // arithmetic, nothing clamps it)
  rptdev::LeafBox lb;
  for (int k = 0; k < 3; k++) {
    const uint32_t c = (q[k] + q[3 + k]) >> 1, h = q[3 + k] - c;
    lb.w[k] = c | (h << 16);
  }
  lb.w[3] = full ? 1u : 0u;
  return lb;
}
void grid_over(const double* bounds, double* qlo, double* qscale) {
  for (int k = 0; k < 3; k++) {
    double ext = bounds[3 + k] - bounds[k];
    qscale[k] = (ext > 0.0 && std::isfinite(ext)) ? ext / 65529.0 : 1.0;
    qlo[k] = bounds[k] - 2.0 * qscale[k];
  }
}
bool sliver(const rptdev::TriX& x) { // ill-conditioned barycentric system, degenerate or NaN: never filtered
  return !(x.denom > 1e-10 * (x.d00 * x.d11)) || !std::isfinite(x.denom);
}

// Spheres (and monomial surfaces) are tested by solving a polynomial whose coefficients grow with the square of the
// origin's distance in OBJECT units: from far away Sphere::intersect accepts lines that miss the sphere (kernels/
// shapes.inc boxray_make).  The device bounds the origin's distance to 1e7 grid steps when quadrics are filtered; a
// sphere whose smallest semi-axis is below 64 steps (1e-3 of the grid) is not filtered at all, nor is a monomial surface.
bool quadric_too_small(const rptdev::Inst& in, const double* qscale) {
  if (in.kind == RPT_SHAPE_MONOMIAL) return true;
  if (in.kind != RPT_SHAPE_SPHERE) return false;
  double r_min = 1.0; // smallest singular value of the placement >= 1 / ||M^-1||_F
  if (in.has_xf) {
    double b = 0.0;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) b += in.inv[4 * c + r] * in.inv[4 * c + r];
    r_min = 1.0 / std::sqrt(b);
  }
  const double step = std::fmax(std::fmax(qscale[0], qscale[1]), qscale[2]);
  return !(r_min >= 64.0 * step);
}

void fill_leaf_boxes(FlatScene& fs, int tree, int64_t tri_base /* < 0: a GROUP tree, entries are placed shapes */,
                     const std::vector<Box>& boxes, const std::vector<rptdev::Inst>* kids = nullptr) {
  rptdev::Tree& t = fs.trees[tree];
  // the grid is two steps larger than the bounds on every side: a coordinate of a primitive maps to [2, 65531], so the
  // outward rounding below (floor - 1, ceil + 1) never reaches the clamp, i.e. a box on a face of the tree's bounds
  // keeps its margin too (tests/test_leaf_boxes.py found the case: a vertex on the bounds, a hit exactly there)
  grid_over(t.bounds, t.qlo, t.qscale);
  size_t nrefs = fs.refs.size() - t.ref_base;
  fs.lbox.resize(fs.refs.size());
  parallel_for(nrefs, 16384, [&](size_t j0, size_t j1) {
  for (size_t j = j0; j < j1; j++) {
    uint32_t tri = fs.refs[t.ref_base + j];
    const Box& b = boxes[tri];
    const bool full = tri_base >= 0 ? sliver(fs.trix[tri_base + tri]) : (kids && quadric_too_small((*kids)[tri], t.qscale));
    rptdev::LeafBox lb = quantise_box(b, t.qlo, t.qscale, full);
    fs.lbox[t.ref_base + j] = lb;
  }
  });
}

// The same filter one level up, for scenes that are a list of many small objects (every tree a single leaf: the flat
// path kernel, kernels/paths.inc flat_query_filtered).  The reference tests every object of scene.objects against every
// ray (renderer.rs:211-220); an object's intersect can only accept a hit point that lies on the object, hence inside its
// bounding_box (the triangles' boxes for a mesh, Transformed::bounding_box shape.rs:153-176 for a placed sphere / cube /
// mesh), so a ray that does not cross that box — enlarged by a grid step, tested in f32 — inside [t_min, record.time]
// skips the object's test with nothing changed.  Never filtered: unbounded objects (Plane), boxes that are not finite,
// meshes with a sliver triangle (see above), placements whose matrices are so ill-conditioned that the world-space
// point o + t d and the object-space point the test accepted could be a noticeable fraction of a grid step apart.
void fill_object_boxes(FlatScene& fs, const std::vector<Box>& world, const std::vector<char>& bounded) {
  const size_t n = world.size();
  fs.obj_filter_ok = n >= 1 && n <= 64;
  fs.obj_always = ~0ull;
  fs.obj_lbox.assign(n, quantise_box(empty_box(), fs.obj_grid, fs.obj_grid + 3, true));
  if (!fs.obj_filter_ok) return;
  std::vector<char> ok(n, 0);
  Box all = empty_box();
  for (size_t i = 0; i < n; i++) {
    const rptdev::Inst& in = fs.insts[i];
    if (in.kind != RPT_SHAPE_SPHERE && in.kind != RPT_SHAPE_CUBE && in.kind != RPT_SHAPE_PLANE && in.kind != RPT_SHAPE_MESH) {
      fs.obj_filter_ok = false; // a kind the filtered walk does not dispatch per lane (group, monomial surface)
      return;
    }
    bool good = bounded[i] != 0;
    for (int k = 0; k < 3 && good; k++) good = std::isfinite(world[i].lo[k]) && std::isfinite(world[i].hi[k]) && world[i].lo[k] <= world[i].hi[k];
    if (good && in.kind == RPT_SHAPE_MESH) {
      const rptdev::Tree& t = fs.trees[in.tree];
      for (uint32_t j = 0; j < t.num_prims && good; j++) good = !sliver(fs.trix[t.prim_base + j]);
    }
    if (good && in.has_xf) { // condition number (Frobenius) of the placement
      double a = 0.0, b = 0.0;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) { a += in.fwd[4 * c + r] * in.fwd[4 * c + r]; b += in.inv[4 * c + r] * in.inv[4 * c + r]; }
      good = std::isfinite(a) && std::isfinite(b) && a * b < 1e8; // cond < 1e4
    }
    ok[i] = good ? 1 : 0;
    if (good) all = merge(all, world[i]);
  }
  double bounds[6];
  for (int k = 0; k < 3; k++) { bounds[k] = all.lo[k]; bounds[3 + k] = all.hi[k]; }
  for (int k = 0; k < 6; k++)
    if (!std::isfinite(bounds[k])) return; // nothing to filter: obj_always stays all ones
  grid_over(bounds, fs.obj_grid, fs.obj_grid + 3);
  std::memcpy(fs.obj_grid + 6, bounds, sizeof(bounds));
  fs.obj_always = 0;
  for (size_t i = 0; i < n; i++) {
    if (ok[i] && quadric_too_small(fs.insts[i], fs.obj_grid + 3)) ok[i] = 0;
    if (ok[i]) fs.obj_lbox[i] = quantise_box(world[i], fs.obj_grid, fs.obj_grid + 3, false);
    if (!ok[i] || fs.obj_lbox[i].w[3]) fs.obj_always |= 1ull << i;
  }
  if (n < 64) fs.obj_always &= (1ull << n) - 1ull;
}

struct Flattener {
  FlatScene& fs;
  std::string& err;
  std::map<std::pair<const void*, uint64_t>, int> mesh_cache; // shared meshes (Arc<Mesh>)
  const BuildOptions* build = nullptr;
  bool light_shape = false; // the shape being flattened is a Light::Object's

  int add_tree(const std::vector<Box>& boxes, uint32_t prim_base) {
    KdBuild kb;
    bool on_device = false;
    if (build && build->device >= 0 && build->device_build_min && boxes.size() >= build->device_build_min) {
      std::string why; // (a refusal is not an error: the host builder makes the same tree)
      on_device = kd_build_device(boxes, kb, build->device, why);
    }
    if (on_device) fs.trees_built_on_device++;
    else kd_build(boxes, kb);
    // (a tree deeper than KD_MAX_STACK is no error: the object it belongs to is walked by rpt_tree_generic, api_scene.cpp)
    fs.max_tree_depth = std::max(fs.max_tree_depth, kb.max_depth);
    rptdev::Tree t;
    std::memset(&t, 0, sizeof(t));
    t.node_base = (uint32_t)fs.nodes.size();
    t.ref_base = (uint32_t)fs.refs.size();
    t.prim_base = prim_base;
    t.num_prims = (uint32_t)boxes.size();
    if (t.num_prims) { // uniform.rs (rand 0.8.3) UniformInt::<usize>::sample: ints_to_reject = (MAX - range + 1) % range
      uint64_t n = t.num_prims;
      t.sample_zone = 0xFFFFFFFFFFFFFFFFull - (0xFFFFFFFFFFFFFFFFull - n + 1) % n;
    }
    t.regular = kb.regular ? 1u : 0u;
    t.split_range_ok = t.regular;
    for (const rptdev::KdNode& nd : kb.nodes) {
      if ((nd.ib & 3u) == 3u) continue;
      const double a = std::fabs(nd.split);
      if (!(a == 0.0 || (a >= 0x1p-340 && a < 0x1p399))) t.split_range_ok = 0u;
    }
    if (!kb.nodes.empty() && (kb.nodes[0].ib & 3u) == 3u) {
      t.root_leaf = 1u + (kb.nodes[0].ib >> 2);
      t.root_first = kb.nodes[0].a;
    }
    Box bounds = empty_box(); // KdTree::new bounds fold kdtree.rs:110-113
    for (const Box& b : boxes) bounds = merge(bounds, b);
    for (int k = 0; k < 3; k++) { t.bounds[k] = bounds.lo[k]; t.bounds[3 + k] = bounds.hi[k]; }
    // child / ref indices stay tree-relative; kernels add node_base / ref_base
    fs.nodes.insert(fs.nodes.end(), kb.nodes.begin(), kb.nodes.end());
    fs.refs.insert(fs.refs.end(), kb.refs.begin(), kb.refs.end());
    fs.trees.push_back(t);
    fs.tree_depth.push_back(kb.max_depth);
    return (int)fs.trees.size() - 1;
  }

  // fills `in` (already placed in fs.insts or a local) from a shape description; returns the
  // untransformed-or-transformed bounding box through *bbox when the shape is Bounded.
  int fill_inst(const RptShape& s, rptdev::Inst& in, Box* bbox, bool* bounded, int nesting) {
    std::memset(&in, 0, sizeof(in));
    in.kind = s.kind;
    in.has_xf = s.transformed ? 1 : 0;
    in.tree = -1;
    in.material = -1;
    if (s.transformed) {
      std::memcpy(in.inv, s.xf.inverse_transform, sizeof(in.inv));
      std::memcpy(in.nrm, s.xf.normal_transform, sizeof(in.nrm));
      std::memcpy(in.fwd, s.xf.transform, sizeof(in.fwd));
      std::memcpy(in.lin, s.xf.linear, sizeof(in.lin));
      in.scale = s.xf.scale;
    }
    Box local = empty_box();
    bool is_bounded = true;
    switch (s.kind) {
      case RPT_SHAPE_SPHERE:
        for (int k = 0; k < 3; k++) { local.lo[k] = -1.0; local.hi[k] = 1.0; }
        break;
      case RPT_SHAPE_CUBE:
        for (int k = 0; k < 3; k++) { local.lo[k] = -0.5; local.hi[k] = 0.5; }
        break;
      case RPT_SHAPE_PLANE:
        for (int k = 0; k < 3; k++) in.plane[k] = s.plane_normal[k];
        in.plane[3] = s.plane_value;
        is_bounded = false;
        break;
      case RPT_SHAPE_MONOMIAL:
        if (nesting > 0) fs.nested_mesh = true; // a tree child of the extended set: the *_ext kernel builds
        if (s.monomial_exp != 4.0) {
          err = "MonomialSurface: intersection and normals are only defined for exp = 4 (monomial_surface.rs:10)";
          return RPTGPU_E_UNSUPPORTED_SHAPE;
        }
        in.plane[0] = s.monomial_height;
        in.plane[1] = s.monomial_exp;
        local.lo[0] = -1.0; local.lo[1] = 0.0; local.lo[2] = -1.0; // monomial_surface.rs:183-190
        local.hi[0] = 1.0; local.hi[1] = s.monomial_height; local.hi[2] = 1.0;
        break;
      case RPT_SHAPE_MESH: {
        if (nesting > 0) fs.nested_mesh = true; // kd-tree of kd-trees (examples/fractal_teapots.rs)
        if (!s.triangles && s.num_triangles) { err = "null triangles"; return RPTGPU_E_INVALID_ARGUMENT; }
        auto key = std::make_pair((const void*)s.triangles, (uint64_t)s.num_triangles);
        auto it = mesh_cache.find(key);
        if (it != mesh_cache.end()) {
          in.tree = it->second;
        } else {
          uint32_t base = (uint32_t)fs.tris.size();
          std::vector<Box> boxes(s.num_triangles);
          fs.tris.resize(base + s.num_triangles);
          fs.trix.resize(base + s.num_triangles);
          parallel_for(s.num_triangles, 16384, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
              const RptTriangle& t = s.triangles[i];
              std::memcpy(fs.tris[base + i].v, &t, sizeof(double) * 18);
              fill_trix(t, fs.trix[base + i]);
              for (int k = 0; k < 3; k++) { // glm::min3 / max3, mesh.rs:40-45
                boxes[i].lo[k] = std::fmin(std::fmin(t.v1[k], t.v2[k]), t.v3[k]);
                boxes[i].hi[k] = std::fmax(std::fmax(t.v1[k], t.v2[k]), t.v3[k]);
              }
            }
          });
          int tr = add_tree(boxes, base);
          if (tr < 0) return RPTGPU_E_TREE_TOO_DEEP;
          // leaf-ordered copies of the intersection records: entry j of refs[] <-> lrec[j], so a
          // leaf's triangles are one contiguous run of 128-byte lines and need no index gather
          // (HBM capacity is spent on locality: refs are ~5x the triangle count, kdtree.rs:270-281)
          {
            const rptdev::Tree& t = fs.trees[tr];
            size_t nrefs = fs.refs.size() - t.ref_base;
            fs.lrec.resize(fs.refs.size());
            parallel_for(nrefs, 16384, [&](size_t j0, size_t j1) {
              for (size_t j = j0; j < j1; j++) fs.lrec[t.ref_base + j] = fs.trix[base + fs.refs[t.ref_base + j]];
            });
          }
          fill_leaf_boxes(fs, tr, (int64_t)base, boxes);
          mesh_cache[key] = tr;
          in.tree = tr;
        }
        const rptdev::Tree& t = fs.trees[in.tree];
        for (int k = 0; k < 3; k++) { local.lo[k] = t.bounds[k]; local.hi[k] = t.bounds[3 + k]; }
        std::memcpy(in.bounds, t.bounds, sizeof(in.bounds));
        break;
      }
      case RPT_SHAPE_GROUP: {
        // KdTree<Box<dyn Bounded>> forwards Bounded through Box (kdtree.rs:14-24), so a group can sit in a group, to
        // any depth: rpt_tree_generic walks such an object (kernels/wavefront.inc).  Only Shape::sample keeps a limit:
        // a LIGHT whose shape nests groups deeper than RPT_MAX_NEST is refused (kernels/sampling.inc sample_child)
        if (light_shape && nesting > RPT_MAX_NEST) {
          err = "Light::Object: KdTree<Box<dyn Bounded>> nested more than " + std::to_string(RPT_MAX_NEST + 1) + " levels deep";
          return RPTGPU_E_UNSUPPORTED_SHAPE;
        }
        if (nesting > 0) fs.nested_mesh = true;
        if (!s.children && s.num_children) { err = "null children"; return RPTGPU_E_INVALID_ARGUMENT; }
        std::vector<rptdev::Inst> kids(s.num_children);
        std::vector<Box> boxes(s.num_children);
        for (uint64_t i = 0; i < s.num_children; i++) {
          const RptShape& c = s.children[i];
          if (c.kind != RPT_SHAPE_SPHERE && c.kind != RPT_SHAPE_CUBE && c.kind != RPT_SHAPE_MESH &&
              c.kind != RPT_SHAPE_MONOMIAL && c.kind != RPT_SHAPE_GROUP) {
            err = "KdTree<Box<dyn Bounded>> children must be Bounded: spheres, cubes, meshes, monomial surfaces or groups "
                  "(optionally Transformed); a Plane is not (kdtree.rs:9-12)";
            return RPTGPU_E_UNSUPPORTED_SHAPE;
          }
          bool b = true;
          int rc = fill_inst(c, kids[i], &boxes[i], &b, nesting + 1);
          if (rc != RPTGPU_OK) return rc;
        }
        uint32_t base = (uint32_t)fs.insts.size();
        int tr = add_tree(boxes, base);
        if (tr < 0) return RPTGPU_E_TREE_TOO_DEEP;
        // the children's boxes (Sphere / Cube / Mesh bounds through Transformed::bounding_box, shape.rs:153-176) contain
        // every point their intersect can return; the same filter as for triangles applies
        fill_leaf_boxes(fs, tr, -1, boxes, &kids);
        for (const rptdev::Inst& k : kids)
          if (k.kind == RPT_SHAPE_MESH) fs.trees[tr].mesh_kids = 1u;
        group_children.push_back({tr, std::move(kids)});
        in.tree = tr;
        const rptdev::Tree& t = fs.trees[tr];
        for (int k = 0; k < 3; k++) { local.lo[k] = t.bounds[k]; local.hi[k] = t.bounds[3 + k]; }
        std::memcpy(in.bounds, t.bounds, sizeof(in.bounds));
        break;
      }
      default:
        err = "unknown shape kind " + std::to_string(s.kind);
        return RPTGPU_E_UNSUPPORTED_SHAPE;
    }
    if (bounded) *bounded = is_bounded;
    if (bbox && is_bounded) *bbox = s.transformed ? transformed_box(local, s.xf.transform) : local;
    return RPTGPU_OK;
  }

  // GROUP children are appended after all top-level / light instances so that
  // insts[0..num_objects) stay the scene's objects; prim_base is patched afterwards.
  std::vector<std::pair<int, std::vector<rptdev::Inst>>> group_children;
};

} // namespace

int flatten_scene(const RptScene& sc, FlatScene& fs, std::string& err, const BuildOptions* build) {
  if ((sc.num_objects && !sc.objects) || (sc.num_lights && !sc.lights)) {
    err = "null objects/lights";
    return RPTGPU_E_INVALID_ARGUMENT;
  }
  struct ThreadsScope { // (restored on every way out)
    int saved;
    explicit ThreadsScope(int v) : saved(t_build_threads) { t_build_threads = v; }
    ~ThreadsScope() { t_build_threads = saved; }
  } threads_scope(build ? build->build_threads : 0);
  Flattener fl{fs, err, {}, build, {}};
  fs.num_objects = (int32_t)sc.num_objects;
  fs.insts.resize(sc.num_objects);
  std::vector<Box> world(sc.num_objects, empty_box()); // Bounded objects: their bounding_box (the object filter's)
  std::vector<char> world_ok(sc.num_objects, 0);
  for (uint64_t i = 0; i < sc.num_objects; i++) {
    rptdev::Inst in;
    bool bounded = false;
    int rc = fl.fill_inst(sc.objects[i].shape, in, &world[i], &bounded, 0);
    if (rc != RPTGPU_OK) return rc;
    world_ok[i] = bounded ? 1 : 0;
    in.material = (int32_t)fs.materials.size();
    fs.insts[i] = in;
    rptdev::Material m;
    std::memset(&m, 0, sizeof(m));
    const RptMaterial& s = sc.objects[i].material;
    { // sample_f's lobe probability (material.rs:233-235) goes to rng.gen_bool(f) (:264), which panics outside
      // [0, 1] (NaN included); the device has no panic, so such a material is refused here
      double f0 = (s.index - 1.0) / (s.index + 1.0);
      f0 = f0 * f0;
      double mean = ((s.color[0] + s.color[1]) + s.color[2]) / 3.0;
      double f = (1.0 - s.metallic) * f0 + s.metallic * mean;
      f = f * (1.0 - 0.2) + 1.0 * 0.2;
      if (!(f >= 0.0 && f <= 1.0)) {
        err = "material of object " + std::to_string(i) + ": specular lobe probability " + std::to_string(f) +
              " is outside [0, 1] (gen_bool would panic, material.rs:264)";
        return RPTGPU_E_INVALID_ARGUMENT;
      }
    }
    std::memcpy(m.color, s.color, sizeof(m.color));
    m.index = s.index; m.roughness = s.roughness; m.metallic = s.metallic;
    m.emittance = s.emittance; m.transparent = s.transparent ? 1 : 0;
    fs.materials.push_back(m);
  }
  for (uint64_t i = 0; i < sc.num_lights; i++) {
    const RptLight& l = sc.lights[i];
    rptdev::Light dl;
    std::memset(&dl, 0, sizeof(dl));
    dl.kind = l.kind;
    dl.inst = -1;
    std::memcpy(dl.color, l.color, sizeof(dl.color));
    std::memcpy(dl.vec, l.vec, sizeof(dl.vec));
    switch (l.kind) {
      case RPT_LIGHT_POINT: case RPT_LIGHT_DIRECTIONAL: fs.num_shadow_lights++; break;
      case RPT_LIGHT_AMBIENT: break;
      case RPT_LIGHT_OBJECT: {
        fs.num_shadow_lights++;
        if (l.object.shape.kind == RPT_SHAPE_PLANE) {
          err = "Light::Object over a Plane: Plane::sample is unimplemented!() (plane.rs:34-36)";
          return RPTGPU_E_UNIMPLEMENTED_SAMPLE;
        }
        rptdev::Inst in;
        fl.light_shape = true;
        int rc = fl.fill_inst(l.object.shape, in, nullptr, nullptr, 0);
        fl.light_shape = false;
        if (rc != RPTGPU_OK) return rc;
        dl.inst = (int32_t)fs.insts.size();
        fs.insts.push_back(in);
        std::memcpy(dl.mat_color, l.object.material.color, sizeof(dl.mat_color));
        dl.mat_emittance = l.object.material.emittance;
        break;
      }
      default: err = "unknown light kind"; return RPTGPU_E_INVALID_ARGUMENT;
    }
    fs.lights.push_back(dl);
  }
  fill_object_boxes(fs, world, world_ok);
  {
    Box all = empty_box();
    bool any = false;
    for (size_t i = 0; i < world.size(); i++) {
      bool good = world_ok[i] != 0;
      for (int k = 0; k < 3 && good; k++) good = std::isfinite(world[i].lo[k]) && std::isfinite(world[i].hi[k]) && world[i].lo[k] <= world[i].hi[k];
      if (good) { all = merge(all, world[i]); any = true; }
    }
    fs.scene_bounds_ok = any;
    for (int k = 0; k < 3 && any; k++) {
      fs.scene_bounds[k] = all.lo[k]; fs.scene_bounds[3 + k] = all.hi[k];
      if (!(all.hi[k] > all.lo[k]) || !std::isfinite(all.hi[k] - all.lo[k])) fs.scene_bounds_ok = false;
    }
  }
  for (auto& g : fl.group_children) { // now place GROUP children and patch prim_base
    fs.trees[g.first].prim_base = (uint32_t)fs.insts.size();
    fs.insts.insert(fs.insts.end(), g.second.begin(), g.second.end());
  }
  // What rpt_tree_generic needs to walk any object of this scene (kernels/wavefront.inc): deferred far children — a
  // tree of depth D defers at most D, plus those of the trees suspended above it — and one frame per tree child entered
  {
    std::vector<int> memo_levels(fs.trees.size(), -1), memo_frames(fs.trees.size(), -1);
    std::function<void(int, bool)> need = [&](int t, bool group) {
      if (memo_levels[t] >= 0) return;
      uint32_t lv = 0, fr = 0;
      fs.tree_kids[t] = 0;
      if (group) {
        const rptdev::Tree& tr = fs.trees[t];
        for (uint32_t k = 0; k < tr.num_prims; k++) {
          const rptdev::Inst& kid = fs.insts[tr.prim_base + k];
          if (kid.kind != RPT_SHAPE_MESH && kid.kind != RPT_SHAPE_GROUP) continue;
          fs.tree_kids[t] |= kid.kind == RPT_SHAPE_GROUP ? 2u : 1u;
          need(kid.tree, kid.kind == RPT_SHAPE_GROUP);
          lv = std::max(lv, (uint32_t)memo_levels[kid.tree]);
          fr = std::max(fr, (uint32_t)memo_frames[kid.tree] + 1u);
          if (fs.tree_kids[kid.tree] & 2u) fs.tree_kids[t] |= 2u;
        }
      }
      memo_levels[t] = (int)(lv + fs.tree_depth[t] + 1u);
      memo_frames[t] = (int)fr;
    };
    fs.tree_kids.assign(fs.trees.size(), 0);
    for (const rptdev::Inst& in : fs.insts)
      if (in.kind == RPT_SHAPE_MESH || in.kind == RPT_SHAPE_GROUP) {
        need(in.tree, in.kind == RPT_SHAPE_GROUP);
        fs.generic_levels = std::max(fs.generic_levels, (uint32_t)memo_levels[in.tree]);
        fs.generic_frames = std::max(fs.generic_frames, (uint32_t)memo_frames[in.tree]);
      }
  }
  fs.lrec.resize(fs.refs.size()); // GROUP trees own ref slots too (unused records)
  // the box batches of the leaf filter read a whole batch from the leaf's first entry on, unclamped — entries past the
  // leaf's end are masked out, not skipped — so the array ends with a batch of padding (kernels/shapes.inc)
  fs.lbox.resize(fs.refs.size() + RPT_LBOX_PAD);
  fs.env_kind = sc.environment.kind;
  std::memcpy(fs.env_color, sc.environment.color, sizeof(fs.env_color));
  if (sc.environment.kind == RPT_ENV_HDRI) {
    if (!sc.environment.texels || !sc.environment.width || !sc.environment.height) {
      err = "HDRI without texels";
      return RPTGPU_E_INVALID_ARGUMENT;
    }
    fs.env_width = sc.environment.width;
    fs.env_height = sc.environment.height;
    size_t n = (size_t)fs.env_width * fs.env_height * 3;
    // two guard rows/texels: the reference reads (x0+1, y0+1) unguarded (environment.rs:41-49)
    fs.env_texels.assign(sc.environment.texels, sc.environment.texels + n);
    fs.env_texels.resize(n + 3 * ((size_t)fs.env_width + 2), 0.0);
  } else if (sc.environment.kind != RPT_ENV_COLOR) {
    err = "unknown environment kind";
    return RPTGPU_E_INVALID_ARGUMENT;
  }
  return RPTGPU_OK;
}

} // namespace rpthost
