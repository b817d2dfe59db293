// rpt.hpp — header-only C++17 host mirror of rpt's builder API over the C ABI (rpt_gpu.h).
//
// The reference is a Rust crate and there is no Rust toolchain in the build image, so the host
// side above the boundary is mirrored here in C++ with the reference's names and argument
// meaning (reference src/lib.rs:9-21 re-exports): Scene / Object / Light / Material / Camera /
// Renderer / Buffer / Filter / sphere() / plane() / cube() / polygon() / Mesh / KdTree /
// hex_color / color_bytes.  This header DESCRIBES scenes and consumes frames; every ray is
// traced by librptgpu.so on the GPU.  Compile with -ffp-contract=off so transforms carry the
// same bits as the other host mirrors.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <istream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rpt_gpu.h"

namespace rpt {

struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};
using Color = Vec3; // src/color.rs:2

struct GpuError : std::runtime_error {
  int code;
  GpuError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

namespace glm { // the nalgebra-glm functions scene construction uses (shape.rs:111-124, 202-284)
using Mat4 = std::vector<double>; // 16, column-major
inline double dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 scale(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 a) {
  double n = std::sqrt(dot(a, a));
  return {a.x / n, a.y / n, a.z / n};
}
inline Mat4 identity4() { return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
inline Mat4 mul4(const Mat4& a, const Mat4& b) { // gemv column accumulation
  Mat4 o(16);
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 4; r++) {
      double acc = a[0 * 4 + r] * b[j * 4 + 0];
      acc = acc + a[1 * 4 + r] * b[j * 4 + 1];
      acc = acc + a[2 * 4 + r] * b[j * 4 + 2];
      acc = acc + a[3 * 4 + r] * b[j * 4 + 3];
      o[j * 4 + r] = acc;
    }
  return o;
}
inline Mat4 translation(Vec3 v) { Mat4 m = identity4(); m[12] = v.x; m[13] = v.y; m[14] = v.z; return m; }
inline Mat4 scaling(Vec3 v) { Mat4 m = identity4(); m[0] = v.x; m[5] = v.y; m[10] = v.z; return m; }
inline Mat4 rotation(double angle, Vec3 axis) { // Rotation3::from_axis_angle(normalize(axis), angle)
  double n = std::sqrt(dot(axis, axis));
  double ux = axis.x / n, uy = axis.y / n, uz = axis.z / n;
  double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
  double s = std::sin(angle), c = std::cos(angle), omc = 1.0 - c;
  double rows[3][3] = {{sqx + (1.0 - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s},
                       {ux * uy * omc + uz * s, sqy + (1.0 - sqy) * c, uy * uz * omc - ux * s},
                       {ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (1.0 - sqz) * c}};
  Mat4 m = identity4();
  for (int r = 0; r < 3; r++)
    for (int col = 0; col < 3; col++) m[col * 4 + r] = rows[r][col];
  return m;
}
inline void mat4_to_mat3(const Mat4& m, double* o) {
  const int idx[9] = {0, 1, 2, 4, 5, 6, 8, 9, 10};
  for (int i = 0; i < 9; i++) o[i] = m[idx[i]];
}
inline double determinant3(const double* m) {
  double m11 = m[0], m21 = m[1], m31 = m[2], m12 = m[3], m22 = m[4], m32 = m[5], m13 = m[6], m23 = m[7], m33 = m[8];
  double a = m22 * m33 - m32 * m23, b = m21 * m33 - m31 * m23, c = m21 * m32 - m31 * m22;
  return m11 * a - m12 * b + m13 * c;
}
inline void inverse_transpose3(const double* m, double* o) { // try_inverse (closed form) then transpose
  double m11 = m[0], m21 = m[1], m31 = m[2], m12 = m[3], m22 = m[4], m32 = m[5], m13 = m[6], m23 = m[7], m33 = m[8];
  double a = m22 * m33 - m32 * m23, b = m21 * m33 - m31 * m23, c = m21 * m32 - m31 * m22;
  double det = m11 * a - m12 * b + m13 * c;
  if (det == 0.0) { std::memset(o, 0, 9 * sizeof(double)); return; }
  double r[3][3];
  r[0][0] = a / det;
  r[0][1] = (m13 * m32 - m33 * m12) / det;
  r[0][2] = (m12 * m23 - m22 * m13) / det;
  r[1][0] = -b / det;
  r[1][1] = (m11 * m33 - m31 * m13) / det;
  r[1][2] = (m13 * m21 - m23 * m11) / det;
  r[2][0] = c / det;
  r[2][1] = (m12 * m31 - m32 * m11) / det;
  r[2][2] = (m11 * m22 - m21 * m12) / det;
  // inverse is r[row][col]; its transpose, stored column-major, is o[col*3+row] = r[col][row]
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 3; row++) o[col * 3 + row] = r[col][row];
}
inline Mat4 inverse4(const Mat4& m) { // nalgebra do_inverse4 (cofactor expansion)
  Mat4 o(16);
  o[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  o[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  o[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  o[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  o[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  o[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  o[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  o[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  o[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  o[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  o[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  o[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  o[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  o[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  o[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  o[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
  if (det == 0.0) return Mat4(16, 0.0);
  double inv_det = 1.0 / det;
  for (double& v : o) v = v * inv_det;
  return o;
}
} // namespace glm

// ---- color.rs:10-24 ----
inline Color hex_color(uint32_t x) {
  double r = ((x >> 16) & 0xff) / 255.0, g = ((x >> 8) & 0xff) / 255.0, b = (x & 0xff) / 255.0;
  return {std::pow(r, 2.2), std::pow(g, 2.2), std::pow(b, 2.2)};
}
inline void color_bytes(const Color& c, uint8_t out[3]) {
  const double v[3] = {c.x, c.y, c.z};
  for (int i = 0; i < 3; i++) {
    double t = std::pow(std::fmin(std::fmax(v[i], 0.0), 1.0), 1.0 / 2.2) * 255.0;
    out[i] = !(t > 0.0) ? 0 : (t >= 255.0 ? 255 : (uint8_t)t); // `as u8`: truncating, NaN -> 0
  }
}

// ---- material.rs:8-105 ----
struct Material {
  Color color = hex_color(0xff0000); // Default: specular(red, 0.5)
  double index = 1.5, roughness = 0.5, metallic = 0.0, emittance = 0.0;
  bool transparent_ = false;
  static Material make(Color c, double i, double r, double m, double e, bool t) {
    Material x; x.color = c; x.index = i; x.roughness = r; x.metallic = m; x.emittance = e; x.transparent_ = t; return x;
  }
  static Material diffuse(Color c) { return make(c, 1.5, 1.0, 0.0, 0.0, false); }
  static Material specular(Color c, double roughness) { return make(c, 1.5, roughness, 0.0, 0.0, false); }
  static Material clear(double index, double roughness) { return make({1, 1, 1}, index, roughness, 0.0, 0.0, true); }
  static Material transparent(Color c, double index, double roughness) { return make(c, index, roughness, 0.0, 0.0, true); }
  static Material metallic_(Color c, double roughness) { return make(c, 1.5, roughness, 1.0, 0.0, false); }
  static Material light(Color c, double emittance) { return make(c, 1.0, 1.0, 0.0, emittance, false); }
  RptMaterial lower() const {
    RptMaterial m{};
    m.color[0] = color.x; m.color[1] = color.y; m.color[2] = color.z;
    m.index = index; m.roughness = roughness; m.metallic = metallic; m.emittance = emittance;
    m.transparent = transparent_ ? 1 : 0;
    return m;
  }
};

// ---- shapes: shape.rs, shape/*.rs, kdtree.rs (description only) ----
struct Triangle { // mesh.rs:8-36
  Vec3 v1, v2, v3, n1, n2, n3;
  static Triangle from_vertices(Vec3 a, Vec3 b, Vec3 c) {
    Vec3 n = glm::normalize(glm::cross(glm::sub(b, a), glm::sub(c, a)));
    return {a, b, c, n, n, n};
  }
};

class Shape;
struct ShapeNode {
  int kind = RPT_SHAPE_SPHERE;
  Vec3 plane_normal;
  double plane_value = 0;
  double monomial_height = 0, monomial_exp = 0;
  std::vector<RptTriangle> triangles;
  std::vector<Shape> children;
};

// keeps every array a lowered scene points into alive
struct Arena {
  std::vector<std::unique_ptr<std::vector<RptShape>>> shapes;
  std::vector<std::shared_ptr<ShapeNode>> nodes;
};

class Shape { // value handle; Transformable (shape.rs:179-284): chained transforms compose as T_new * M_old
 public:
  std::shared_ptr<ShapeNode> node;
  bool transformed = false;
  glm::Mat4 M = glm::identity4();

  Shape with(const glm::Mat4& t) const {
    Shape s = *this;
    s.M = transformed ? glm::mul4(t, M) : t;
    s.transformed = true;
    return s;
  }
  Shape translate(Vec3 v) const { return with(glm::translation(v)); }
  Shape scale(Vec3 v) const { return with(glm::scaling(v)); }
  Shape rotate(double angle, Vec3 axis) const { return with(glm::rotation(angle, axis)); }
  Shape rotate_x(double a) const { return rotate(a, {1, 0, 0}); }
  Shape rotate_y(double a) const { return rotate(a, {0, 1, 0}); }
  Shape rotate_z(double a) const { return rotate(a, {0, 0, 1}); }
  Shape transform(const glm::Mat4& m) const { return with(m); }

  RptShape lower(Arena& arena) const {
    RptShape s{};
    s.kind = node->kind;
    arena.nodes.push_back(node);
    if (transformed) { // Transformed::new, shape.rs:111-124
      s.transformed = 1;
      glm::Mat4 inv = glm::inverse4(M);
      std::memcpy(s.xf.transform, M.data(), sizeof s.xf.transform);
      std::memcpy(s.xf.inverse_transform, inv.data(), sizeof s.xf.inverse_transform);
      glm::mat4_to_mat3(M, s.xf.linear);
      s.xf.scale = glm::determinant3(s.xf.linear);
      glm::inverse_transpose3(s.xf.linear, s.xf.normal_transform);
    }
    s.plane_normal[0] = node->plane_normal.x; s.plane_normal[1] = node->plane_normal.y; s.plane_normal[2] = node->plane_normal.z;
    s.plane_value = node->plane_value;
    s.monomial_height = node->monomial_height;
    s.monomial_exp = node->monomial_exp;
    if (node->kind == RPT_SHAPE_MESH) {
      s.triangles = node->triangles.data();
      s.num_triangles = node->triangles.size();
    } else if (node->kind == RPT_SHAPE_GROUP) {
      auto arr = std::make_unique<std::vector<RptShape>>();
      for (const Shape& c : node->children) arr->push_back(c.lower(arena));
      s.children = arr->data();
      s.num_children = arr->size();
      arena.shapes.push_back(std::move(arr));
    }
    return s;
  }
};

inline Shape make_shape(int kind) { Shape s; s.node = std::make_shared<ShapeNode>(); s.node->kind = kind; return s; }
inline Shape sphere() { return make_shape(RPT_SHAPE_SPHERE); }   // shape.rs:287-289
inline Shape cube() { return make_shape(RPT_SHAPE_CUBE); }       // shape.rs:302-304
inline Shape plane(Vec3 normal, double value) {                  // shape.rs:297-299
  Shape s = make_shape(RPT_SHAPE_PLANE);
  s.node->plane_normal = normal;
  s.node->plane_value = value;
  return s;
}
inline Shape monomial_surface(double height, double exp) {        // shape.rs:292-294
  Shape s = make_shape(RPT_SHAPE_MONOMIAL);
  s.node->monomial_height = height;
  s.node->monomial_exp = exp;
  return s;
}
inline Shape Mesh(const std::vector<Triangle>& tris) { // Mesh = KdTree<Triangle>, mesh.rs:102
  Shape s = make_shape(RPT_SHAPE_MESH);
  for (const Triangle& t : tris) {
    RptTriangle r;
    const Vec3* src[6] = {&t.v1, &t.v2, &t.v3, &t.n1, &t.n2, &t.n3};
    double* dst[6] = {r.v1, r.v2, r.v3, r.n1, r.n2, r.n3};
    for (int i = 0; i < 6; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    s.node->triangles.push_back(r);
  }
  return s;
}
inline Shape KdTree(const std::vector<Shape>& objects) { // KdTree<Box<dyn Bounded>>
  Shape s = make_shape(RPT_SHAPE_GROUP);
  s.node->children = objects;
  return s;
}
inline Shape polygon(const std::vector<Vec3>& verts) { // shape.rs:307-313
  std::vector<Triangle> tris;
  for (size_t i = 1; i + 1 < verts.size(); i++) tris.push_back(Triangle::from_vertices(verts[0], verts[i], verts[i + 1]));
  return Mesh(tris);
}

// ---------------------------------------------------------------------------- asset import (src/io.rs)
// Same parsing rules as the reference: fan triangulation (io.rs:181-198), negative indices count from the
// end (io.rs:10-18), a corner without a normal index makes the whole triangle flat (io.rs:186-187); STL is
// binary when size == 84 + 50 n, else ASCII when it starts with "solid " (io.rs:260-287).
namespace io_detail {
inline std::vector<std::string> tokens(const std::string& line) {
  std::vector<std::string> t;
  std::istringstream ss(line);
  std::string w;
  while (ss >> w) t.push_back(w);
  return t;
}
inline double number(const std::string& s) {
  char* end = nullptr;
  double v = std::strtod(s.c_str(), &end);
  if (end == s.c_str() || *end != '\0') throw std::runtime_error("Failed to parse number in asset file: " + s);
  return v;
}
inline bool parse_index(const std::string& value, size_t len, size_t& out) { // io.rs:10-18
  if (value.empty()) return false;
  char* end = nullptr;
  long long index = std::strtoll(value.c_str(), &end, 10);
  if (end == value.c_str() || *end != '\0') return false;
  out = index > 0 ? (size_t)(index - 1) : (size_t)((long long)len + index);
  return true;
}
inline Vec3 point(const std::vector<std::string>& t) { // parse_obj_point io.rs:150-161
  if (t.size() < 4) throw std::runtime_error("Failed to parse vertex in .OBJ");
  return {number(t[1]), number(t[2]), number(t[3])};
}
inline void face(const std::vector<std::string>& t, const std::vector<Vec3>& vertices, const std::vector<Vec3>& normals,
                 std::vector<Triangle>& out) { // parse_obj_face io.rs:163-200
  std::vector<size_t> vi, vni;
  std::vector<bool> has_n;
  for (size_t k = 1; k < t.size(); k++) {
    std::string args[3];
    size_t a = 0;
    for (char ch : t[k]) {
      if (ch == '/') { if (++a > 2) break; }
      else args[a] += ch;
    }
    size_t v = 0, n = 0;
    if (!parse_index(args[0], vertices.size(), v)) throw std::runtime_error("Invalid vertex index");
    vi.push_back(v);
    has_n.push_back(parse_index(args[2], normals.size(), n));
    vni.push_back(n);
  }
  for (size_t i = 1; i + 1 < vi.size(); i++) {
    const size_t a = 0, b = i, c = i + 1;
    if (!has_n[a] || !has_n[b] || !has_n[c]) {
      out.push_back(Triangle::from_vertices(vertices.at(vi[a]), vertices.at(vi[b]), vertices.at(vi[c])));
    } else {
      out.push_back(Triangle{vertices.at(vi[a]), vertices.at(vi[b]), vertices.at(vi[c]), normals.at(vni[a]),
                             normals.at(vni[b]), normals.at(vni[c])});
    }
  }
}
} // namespace io_detail

inline Shape load_obj(std::istream& in) { // io.rs:27-74
  std::vector<Vec3> vertices, normals;
  std::vector<Triangle> triangles;
  std::string line;
  while (std::getline(in, line)) {
    std::vector<std::string> t = io_detail::tokens(line);
    if (t.empty() || t[0][0] == '#') continue;
    if (t[0] == "v") vertices.push_back(io_detail::point(t));
    else if (t[0] == "vn") normals.push_back(io_detail::point(t));
    else if (t[0] == "f") io_detail::face(t, vertices, normals, triangles);
  }
  return Mesh(triangles);
}
inline Shape load_obj(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  return load_obj(f);
}

inline Shape load_stl(const std::string& path) { // io.rs:260-360
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const size_t size = data.size();
  if (size < 15) throw std::runtime_error("Loaded .STL file is too short");
  std::vector<Triangle> tris;
  if (size >= 84) {
    uint32_t n = 0;
    std::memcpy(&n, data.data() + 80, 4);
    if (size == 84 + (size_t)n * 50) { // very likely binary
      for (uint32_t i = 0; i < n; i++) {
        float v[12];
        std::memcpy(v, data.data() + 84 + (size_t)i * 50, sizeof v);
        Vec3 vn{(double)v[0], (double)v[1], (double)v[2]}; // f32 -> f64, as the reference does
        tris.push_back(Triangle{{(double)v[3], (double)v[4], (double)v[5]}, {(double)v[6], (double)v[7], (double)v[8]},
                                {(double)v[9], (double)v[10], (double)v[11]}, vn, vn, vn});
      }
      return Mesh(tris);
    }
  }
  if (data.compare(0, 6, "solid ") == 0) {
    std::vector<std::string> lines;
    std::istringstream ss(data);
    std::string line;
    while (std::getline(ss, line)) lines.push_back(line);
    auto trim = [](const std::string& s) {
      size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
      return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    for (size_t i = 1; i < lines.size();) {
      std::string l = trim(lines[i]);
      if (l.compare(0, 13, "facet normal ") != 0) {
        if (l.empty() || l.compare(0, 8, "endsolid") == 0) { i++; continue; }
        throw std::runtime_error("Malformed STL file: expected `facet normal`");
      }
      std::vector<std::string> nt = io_detail::tokens(l.substr(13));
      if (nt.size() < 3 || i + 4 >= lines.size()) throw std::runtime_error("Malformed STL file");
      Vec3 vn{io_detail::number(nt[0]), io_detail::number(nt[1]), io_detail::number(nt[2])};
      Vec3 vs[3];
      for (int j = 0; j < 3; j++) {
        std::string vl = trim(lines[i + 2 + j]);
        if (vl.compare(0, 7, "vertex ") != 0) throw std::runtime_error("Malformed STL file: expected `vertex`");
        std::vector<std::string> vt = io_detail::tokens(vl.substr(7));
        if (vt.size() < 3) throw std::runtime_error("Malformed STL file");
        vs[j] = {io_detail::number(vt[0]), io_detail::number(vt[1]), io_detail::number(vt[2])};
      }
      tris.push_back(Triangle{vs[0], vs[1], vs[2], vn, vn, vn});
      i += 7;
    }
    return Mesh(tris);
  }
  throw std::runtime_error("Loaded .STL file, but could not determine format");
}

// ---- object.rs:10-31 ----
struct Object {
  Shape shape;
  Material material_;
  explicit Object(Shape s) : shape(std::move(s)) {}
  Object material(const Material& m) const { Object o = *this; o.material_ = m; return o; }
};

// ---- io.rs:202-254 load_mtl, io.rs:83-148 load_obj_with_mtl ----
inline std::map<std::string, Material> load_mtl(std::istream& in) {
  std::map<std::string, Material> materials;
  std::string current, line;
  bool have = false;
  while (std::getline(in, line)) {
    std::vector<std::string> t = io_detail::tokens(line);
    if (t.empty() || t[0][0] == '#') continue;
    if (t[0] == "newmtl") {
      if (t.size() < 2) throw std::runtime_error("newmtl without a name");
      current = t[1];
      have = true;
      materials.emplace(current, Material()); // entry().or_default()
    } else {
      if (!have) throw std::runtime_error("Material was not specified with `newmtl` before properties were added");
      Material& mat = materials[current];
      // best-effort conversion from Ka/Kd/Ks to the physically-based material (io.rs:226-251)
      if (t[0] == "Kd") {
        mat.color = io_detail::point(t);
      } else if (t[0] == "Ns") {
        if (t.size() < 2) throw std::runtime_error("Could not parse Ks value");
        mat.roughness = std::sqrt(std::sqrt(2.0 / (io_detail::number(t[1]) + 2.0)));
      } else if (t[0] == "Ni") {
        if (t.size() < 2) throw std::runtime_error("Could not parse Ns value");
        mat.index = std::fmax(io_detail::number(t[1]), 1.0 + 1e-4); // an IOR of exactly 1.0 cannot be handled
      } else if (t[0] == "d") {
        if (t.size() < 2) throw std::runtime_error("Could not parse d value");
        if (io_detail::number(t[1]) < 0.8) mat.transparent_ = true;
      }
    }
  }
  return materials;
}

// one Object per run of faces under one `usemtl` (a mesh with several materials is several objects, io.rs:122-147)
inline std::vector<Object> load_obj_with_mtl(std::istream& obj, std::istream& mtl) {
  std::map<std::string, Material> materials = load_mtl(mtl);
  std::vector<Vec3> vertices, normals;
  std::vector<Object> objects;
  std::vector<Triangle> current_triangles;
  Material current_material;
  std::string last_usemtl, line;
  bool have_last = false;
  auto flush = [&]() {
    if (!current_triangles.empty()) {
      objects.push_back(Object(Mesh(current_triangles)).material(current_material));
      current_triangles.clear();
    }
  };
  while (std::getline(obj, line)) {
    std::vector<std::string> t = io_detail::tokens(line);
    if (t.empty() || t[0][0] == '#') continue;
    if (t[0] == "v") vertices.push_back(io_detail::point(t));
    else if (t[0] == "vn") normals.push_back(io_detail::point(t));
    else if (t[0] == "f") io_detail::face(t, vertices, normals, current_triangles);
    else if (t[0] == "usemtl") {
      if (t.size() < 2) throw std::runtime_error("usemtl without a name");
      if (!have_last || last_usemtl != t[1]) {
        flush();
        auto it = materials.find(t[1]);
        if (it == materials.end()) throw std::runtime_error("Could not found `usemtl " + t[1] + "` in library");
        current_material = it->second;
        last_usemtl = t[1];
        have_last = true;
      }
    }
  }
  flush();
  return objects;
}
inline std::vector<Object> load_obj_with_mtl(const std::string& obj_path, const std::string& mtl_path) {
  std::ifstream o(obj_path), m(mtl_path);
  if (!o) throw std::runtime_error("cannot open " + obj_path);
  if (!m) throw std::runtime_error("cannot open " + mtl_path);
  return load_obj_with_mtl(o, m);
}

// ---- light.rs:7-19 ----
struct Light {
  int kind = RPT_LIGHT_AMBIENT;
  Color color;
  Vec3 vec;
  std::shared_ptr<rpt::Object> object;
  static Light Point(Color c, Vec3 location) { Light l; l.kind = RPT_LIGHT_POINT; l.color = c; l.vec = location; return l; }
  static Light Ambient(Color c) { Light l; l.kind = RPT_LIGHT_AMBIENT; l.color = c; return l; }
  static Light Directional(Color c, Vec3 dir) { Light l; l.kind = RPT_LIGHT_DIRECTIONAL; l.color = c; l.vec = dir; return l; }
  static Light Object(const rpt::Object& o) { Light l; l.kind = RPT_LIGHT_OBJECT; l.object = std::make_shared<rpt::Object>(o); return l; }
};

// ---- environment.rs:5-78 ----
struct Hdri {
  uint32_t width = 0, height = 0;
  std::vector<double> buf; // width*height*3
};
struct Environment {
  Color color;
  std::shared_ptr<Hdri> hdri;
  static Environment Color_(Color c) { Environment e; e.color = c; return e; }
  static Environment Hdri_(const Hdri& h) { Environment e; e.hdri = std::make_shared<Hdri>(h); return e; }
};

// ---- scene.rs:7-41 ----
struct Scene {
  std::vector<Object> objects;
  std::vector<Light> lights;
  Environment environment;
  void add(const Object& o) { objects.push_back(o); }
  void add(const Light& l) { lights.push_back(l); }
};

// ---- camera.rs:8-62 ----
struct Camera {
  Vec3 eye{0, 0, 10}, direction{0, 0, -1}, up{0, 1, 0};
  double fov = 3.14159265358979323846 / 6.0, aperture = 0.0, focal_distance = 0.0;
  static Camera look_at(Vec3 eye, Vec3 center, Vec3 up, double fov) {
    Camera c;
    c.eye = eye;
    c.direction = glm::normalize(glm::sub(center, eye));
    c.up = glm::normalize(glm::sub(up, glm::scale(c.direction, glm::dot(up, c.direction))));
    c.fov = fov;
    return c;
  }
  Camera focus(Vec3 focal_point, double aperture_) const {
    Camera c = *this;
    c.focal_distance = glm::dot(glm::sub(focal_point, eye), direction);
    c.aperture = aperture_;
    return c;
  }
  RptCamera lower() const {
    RptCamera c{};
    c.eye[0] = eye.x; c.eye[1] = eye.y; c.eye[2] = eye.z;
    c.direction[0] = direction.x; c.direction[1] = direction.y; c.direction[2] = direction.z;
    c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
    c.fov = fov; c.aperture = aperture; c.focal_distance = focal_distance;
    return c;
  }
};

// ---- buffer.rs ----
struct Filter {
  uint32_t radius = 0;
  static Filter Box(uint32_t r) { Filter f; f.radius = r; return f; }
};
struct RgbImage {
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> data; // H*W*3
  void save_ppm(const std::string& path) const {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
  }
};
class Buffer {
 public:
  uint32_t width, height;
  Filter filter;
  std::vector<std::vector<double>> samples; // one W*H*3 array per add_samples call
  Buffer(uint32_t w, uint32_t h, Filter f = Filter()) : width(w), height(h), filter(f) {}
  void add_samples(const std::vector<double>& s) { // buffer.rs:32-40
    if (s.size() != (size_t)width * height * 3) throw std::runtime_error("Invalid sample dimension");
    samples.push_back(s);
  }
  RgbImage image() const { // buffer.rs:43-56 + get_filtered_color :75-93
    RgbImage img;
    img.width = width; img.height = height;
    img.data.resize((size_t)width * height * 3);
    uint32_t r = filter.radius;
    for (uint32_t y = 0; y < height; y++)
      for (uint32_t x = 0; x < width; x++) {
        double c[3] = {0, 0, 0};
        uint64_t count = 0;
        for (uint32_t i = x >= r ? x - r : 0; i <= x + r; i++)
          for (uint32_t j = y >= r ? y - r : 0; j <= y + r; j++)
            if (i < width && j < height) {
              double sum[3] = {0, 0, 0};
              size_t idx = ((size_t)j * width + i) * 3;
              for (const auto& b : samples)
                for (int k = 0; k < 3; k++) sum[k] = sum[k] + b[idx + k];
              for (int k = 0; k < 3; k++) c[k] = c[k] + sum[k];
              count += samples.size();
            }
        if (!count) throw std::runtime_error("Pixel found with no samples");
        color_bytes({c[0] / (double)count, c[1] / (double)count, c[2] / (double)count}, &img.data[((size_t)y * width + x) * 3]);
      }
    return img;
  }
  double variance() const { // buffer.rs:59-73
    double variance = 0.0, count = 0.0, n = (double)samples.size();
    for (size_t p = 0; p < (size_t)width * height; p++) {
      double sum[3] = {0, 0, 0};
      for (const auto& b : samples)
        for (int k = 0; k < 3; k++) sum[k] = sum[k] + b[p * 3 + k];
      double ss = 0.0;
      for (const auto& b : samples) {
        double d0 = b[p * 3] - sum[0] / n, d1 = b[p * 3 + 1] - sum[1] / n, d2 = b[p * 3 + 2] - sum[2] / n;
        ss += (d0 * d0 + d1 * d1) + d2 * d2;
      }
      variance += ss / (n - 1.0);
      count += 1.0;
    }
    return variance / count;
  }
};

// ---- renderer.rs:18-129 ----
class Renderer {
 public:
  Renderer(const Scene& scene, const Camera& camera) : scene_(scene), camera_(camera) {}
  ~Renderer() { if (handle_) rptgpu_scene_destroy(handle_); }
  Renderer(const Renderer&) = delete;
  Renderer& width(uint32_t w) { width_ = w; return *this; }
  Renderer& height(uint32_t h) { height_ = h; return *this; }
  Renderer& exposure_value(double ev) { ev_ = ev; return *this; }
  Renderer& filter(Filter f) { filter_ = f; return *this; }
  Renderer& max_bounces(uint32_t b) { max_bounces_ = b; return *this; }
  Renderer& num_samples(uint32_t n) { num_samples_ = n; return *this; }
  Renderer& seed(uint64_t s) { seed_ = s; return *this; }   // addition: the reference seeds from entropy
  Renderer& device(int d) { device_ = d; return *this; }
  // addition: the back-end's knobs (RptSceneOptions, include/rpt_gpu.h; start from rptgpu_scene_options_default)
  Renderer& gpu_options(const RptSceneOptions& o) { options_ = o; have_options_ = true; return *this; }

  RgbImage render() { // renderer.rs:96-100
    Buffer buffer(width_, height_, filter_);
    samples_done_ = 0;
    sample(num_samples_, buffer);
    return buffer.image();
  }
  void iterative_render(uint32_t callback_interval, const std::function<void(uint32_t, const Buffer&)>& callback) {
    Buffer buffer(width_, height_, filter_); // renderer.rs:103-115
    uint32_t iteration = 0;
    samples_done_ = 0;
    while (iteration < num_samples_) {
      uint32_t steps = std::min(num_samples_ - iteration, callback_interval);
      sample(steps, buffer);
      iteration += steps;
      callback(iteration, buffer);
    }
  }
  // renderer.rs:117-129: the hot path, one call into the C ABI
  void sample(uint32_t iterations, Buffer& buffer) {
    ensure_scene();
    RptRenderParams p{};
    p.width = width_; p.height = height_; p.max_bounces = max_bounces_; p.iterations = iterations;
    p.exposure_value = ev_; p.seed = seed_; p.sample_index_base = samples_done_;
    p.tile_width = 32; p.tile_height = 8; p.part_index = 0; p.part_count = 1;
    RptCamera cam = camera_.lower();
    std::vector<double> colors((size_t)width_ * height_ * 3);
    check(rptgpu_render_batch(handle_, &cam, &p, colors.data()));
    samples_done_ += iterations;
    buffer.add_samples(colors);
  }

 private:
  void check(int code) {
    if (code != RPTGPU_OK) {
      std::string msg = rptgpu_strerror(code);
      const char* d = rptgpu_last_error_detail(handle_);
      if (d && *d) msg += std::string(" - ") + d;
      throw GpuError(code, msg);
    }
  }
  void ensure_scene() {
    if (handle_) return;
    Arena arena;
    std::vector<RptObject> objs;
    for (const Object& o : scene_.objects) objs.push_back({o.shape.lower(arena), o.material_.lower()});
    std::vector<RptLight> lights;
    for (const Light& l : scene_.lights) {
      RptLight r{};
      r.kind = l.kind;
      r.color[0] = l.color.x; r.color[1] = l.color.y; r.color[2] = l.color.z;
      r.vec[0] = l.vec.x; r.vec[1] = l.vec.y; r.vec[2] = l.vec.z;
      if (l.kind == RPT_LIGHT_OBJECT) r.object = {l.object->shape.lower(arena), l.object->material_.lower()};
      lights.push_back(r);
    }
    RptScene s{};
    s.objects = objs.data(); s.num_objects = objs.size();
    s.lights = lights.data(); s.num_lights = lights.size();
    const Environment& e = scene_.environment;
    s.environment.color[0] = e.color.x; s.environment.color[1] = e.color.y; s.environment.color[2] = e.color.z;
    if (e.hdri) {
      s.environment.kind = RPT_ENV_HDRI;
      s.environment.width = e.hdri->width; s.environment.height = e.hdri->height;
      s.environment.texels = e.hdri->buf.data();
    }
    check(have_options_ ? rptgpu_scene_create_opts(&s, device_, &options_, &handle_) : rptgpu_scene_create(&s, device_, &handle_));
  }
  const Scene& scene_;
  Camera camera_;
  uint32_t width_ = 800, height_ = 600, max_bounces_ = 0, num_samples_ = 1; // renderer.rs:46-57
  double ev_ = 0.0;
  Filter filter_;
  uint64_t seed_ = 0x52505447, samples_done_ = 0;
  int device_ = 0;
  RptSceneOptions options_{};
  bool have_options_ = false;
  rptgpu_scene* handle_ = nullptr;
};

} // namespace rpt
