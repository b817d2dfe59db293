// host_scene.h — host side of rptgpu_scene_create: kd-tree construction by the reference rule
// and flattening of the boundary's POD scene description into the device layout.
#pragma once
#include <string>
#include <vector>

#include "../../include/rpt_gpu.h"
#include "device_types.h"

namespace rpthost {

struct Box {
  double lo[3], hi[3];
};

struct KdBuild {
  std::vector<rptdev::KdNode> nodes;
  std::vector<uint32_t> refs;
  uint32_t max_depth = 0;
  bool regular = true; // every split plane lies inside its node's cell
};

// KdTree::new (kdtree.rs:108-119) -> construct (kdtree.rs:235-345), flattened.
// threads <= 0: std::thread::hardware_concurrency() (RPTGPU_BUILD_THREADS overrides), at most 32; the tree is the same
// for any thread count (host_scene.cpp, splice)
void kd_build(const std::vector<Box>& boxes, KdBuild& out, int threads = 0);

// The same tree, node for node, built on HIP device `device` (kdbuild.hip: events sorted once per axis, then one round
// of scans and stable scatters per tree level).  Returns false — `why` says why — when it cannot be used (fewer than 16
// primitives, non-finite boxes, a HIP error, a tree deeper than the device stack): the caller builds on the host then.
bool kd_build_device(const std::vector<Box>& boxes, KdBuild& out, int device, std::string& why);

// scene hand-off options: trees of at least device_build_min primitives are built on the device (0: never)
struct BuildOptions {
  int device = -1;
  size_t device_build_min = 0;
  int build_threads = 0; // RptSceneOptions::build_threads: 0 = the usable cores
};

struct FlatScene {
  std::vector<rptdev::Inst> insts;
  std::vector<rptdev::Tree> trees;
  std::vector<uint32_t> tree_depth; // per tree: depth of its deepest leaf
  std::vector<uint8_t> tree_kids;   // per tree (GROUP): bit 0 a child is a MESH, bit 1 a GROUP sits somewhere below it
  uint32_t generic_levels = 0, generic_frames = 0; // heights of rpt_tree_generic's columns that hold any object of the scene
  std::vector<rptdev::KdNode> nodes;
  std::vector<uint32_t> refs;
  std::vector<rptdev::Tri> tris;
  std::vector<rptdev::TriX> trix; // by triangle index (host-side staging only)
  std::vector<rptdev::TriX> lrec; // by refs[] position: what the device reads
  std::vector<rptdev::LeafBox> lbox; // by refs[] position: conservative boxes in front of the exact test
  std::vector<rptdev::Material> materials;
  std::vector<rptdev::Light> lights;
  std::vector<double> env_texels;
  double env_color[3] = {0, 0, 0};
  uint32_t env_width = 0, env_height = 0;
  int32_t env_kind = 0;
  int32_t num_objects = 0;
  int32_t num_shadow_lights = 0;
  uint32_t max_tree_depth = 0;
  // flat scenes' object filter (kernels/paths.inc flat_query_filtered): one conservative box per TOP-LEVEL object on a
  // grid over all bounded ones; bit k of obj_always: object k is never filtered.  obj_filter_ok: every object is of a
  // kind the filtered walk handles (sphere, cube, plane, mesh) and there are at most 64 of them
  std::vector<rptdev::LeafBox> obj_lbox;
  double obj_grid[12] = {0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0}; // qlo[3], qscale[3], bounds[6]
  uint64_t obj_always = ~0ull;
  bool obj_filter_ok = false;
  // union of the world-space boxes of the bounded top-level objects (planes have none): the grid of the path re-order's
  // sort key (kernels/wavefront.inc rpt_path_keys); ok: there is at least one and it is finite and not flat
  double scene_bounds[6] = {0, 0, 0, 1, 1, 1};
  bool scene_bounds_ok = false;
  uint32_t trees_built_on_device = 0; // how many of the trees kd_build_device made (diagnostics)
  bool nested_mesh = false; // some KdTree<Box<dyn Bounded>> child is a Mesh, a MonomialSurface or another group:
                            // the scene needs the extended kernel builds
};

// returns RPTGPU_OK or an error code; `err` explains
int flatten_scene(const RptScene& scene, FlatScene& out, std::string& err, const BuildOptions* build = nullptr);

} // namespace rpthost
