// Host-side check of the object filter's tables (rpt_amd/csrc/host_scene.cpp fill_object_boxes): builds a scene with the
// C++ mirror's builders, flattens it with the PRODUCT's flattener (no GPU), and prints which objects are exempt from
// the filter and every object's decoded box next to its bounding box.  Driven by tests/test_object_boxes_host.py.
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rpt.hpp"
#include "../../rpt_amd/csrc/host_scene.h"

namespace rpthost { // the device builder is not linked into this test: the host builder makes the same tree
bool kd_build_device(const std::vector<Box>&, KdBuild&, int, std::string& why) { why = "not linked"; return false; }
}

int main() {
  using namespace rpt;
  std::vector<Shape> shapes;
  shapes.push_back(plane({0.0, 1.0, 0.0}, -1.0));                                             // 0 unbounded
  shapes.push_back(sphere().scale({0.5, 0.5, 0.5}).translate({1.0, 0.0, 0.0}));             // 1 filtered
  shapes.push_back(cube().rotate_y(0.7).scale({0.5, 1.5, 0.5}).translate({2.5, 0.0, -2.5})); // 2 filtered
  shapes.push_back(polygon({{-2.0, 0.0, -2.0}, {-2.0, 0.0, 2.0}, {2.0, 0.0, 2.0}, {2.0, 0.0, -2.0}})); // 3 filtered
  shapes.push_back(polygon({{0.0, 1.5, 0.0}, {0.0, 1.5, 0.0}, {1.0, 1.5, 0.0}, {1.0, 1.5000000001, 1e-9}})); // 4 sliver
  shapes.push_back(sphere().scale({1.0, 1e-5, 1.0}).translate({-1.0, 0.5, 0.0}));           // 5 condition number 1e5
  shapes.push_back(sphere().scale({2e-3, 2e-3, 2e-3}).translate({0.2, 0.3, 1.0}));          // 6 below 64 grid steps
  shapes.push_back(sphere().translate({3.0, 2.0, 3.0}));                                      // 7 filtered, untransformed-sized
  shapes.push_back(polygon({{-3.0, 0.0, -3.0}, {-3.0, 2.0, -3.0}, {-3.0, 2.0, 3.0}}).rotate_y(0.3)); // 8 transformed mesh
  Arena arena;
  std::vector<RptObject> objs;
  for (const Shape& s : shapes) objs.push_back({s.lower(arena), Material::diffuse({0.5, 0.5, 0.5}).lower()});
  RptScene sc{};
  sc.objects = objs.data();
  sc.num_objects = objs.size();
  rpthost::FlatScene fs;
  std::string err;
  int rc = rpthost::flatten_scene(sc, fs, err, nullptr);
  std::printf("rc %d ok %d n %zu always %llx\n", rc, fs.obj_filter_ok ? 1 : 0, fs.obj_lbox.size(), (unsigned long long)fs.obj_always);
  std::printf("grid %.17g %.17g %.17g %.17g %.17g %.17g\n", fs.obj_grid[0], fs.obj_grid[1], fs.obj_grid[2], fs.obj_grid[3],
              fs.obj_grid[4], fs.obj_grid[5]);
  for (size_t i = 0; i < fs.obj_lbox.size(); i++) {
    const rptdev::LeafBox& b = fs.obj_lbox[i];
    int q[6]; // decoded as the device does: centre -+ half-extent per axis (device_types.h LeafBox)
    for (int k = 0; k < 3; k++) {
      const int c = (int)(b.w[k] & 0xffffu), e = (int)(b.w[k] >> 16);
      q[k] = c - e; q[3 + k] = c + e;
    }
    std::printf("box %zu full %u q %d %d %d %d %d %d\n", i, b.w[3], q[0], q[1], q[2], q[3], q[4], q[5]);
  }
  // a group among the objects: the filtered walk does not dispatch it, the scene is not filtered at all
  std::vector<Shape> kids = {sphere(), cube().translate({2.0, 0.0, 0.0})};
  objs.push_back({KdTree(kids).lower(arena), Material::diffuse({0.5, 0.5, 0.5}).lower()});
  sc.objects = objs.data();
  sc.num_objects = objs.size();
  rpthost::FlatScene fs2;
  rc = rpthost::flatten_scene(sc, fs2, err, nullptr);
  std::printf("with_group rc %d ok %d\n", rc, fs2.obj_filter_ok ? 1 : 0);
  // the same rule one level down (fill_leaf_boxes for GROUP trees): a child sphere below 64 steps of the group's grid and
  // a monomial surface carry the whole grid, their neighbours a box
  {
    std::vector<Shape> ring;
    for (int i = 0; i < 20; i++) ring.push_back(sphere().scale({0.3, 0.3, 0.3}).translate({0.7 * i, 0.0, 0.0}));
    ring.push_back(sphere().scale({1e-4, 1e-4, 1e-4}).translate({3.0, 1.0, 0.0}));   // child 20
    ring.push_back(monomial_surface(0.5, 4.0).translate({5.0, 1.0, 0.0}));            // child 21
    std::vector<RptObject> o2 = {{KdTree(ring).lower(arena), Material::diffuse({0.5, 0.5, 0.5}).lower()}};
    RptScene s2{};
    s2.objects = o2.data();
    s2.num_objects = o2.size();
    rpthost::FlatScene f3;
    rc = rpthost::flatten_scene(s2, f3, err, nullptr);
    std::printf("group rc %d trees %zu\n", rc, f3.trees.size());
    const rptdev::Tree& t = f3.trees[0];
    for (size_t j = t.ref_base; j < f3.refs.size(); j++) std::printf("ref child %u full %u\n", f3.refs[j], f3.lbox[j].w[3]);
  }
  return 0;
}
