// api_scene.cpp — RptSceneOptions (defaults, sized access, validation, environment overrides), rptgpu_scene_create[_opts]
// (flattening, kd builds, routing between the pipelines, upload), the kd-tree entry points (see api_internal.h)
#include "api_internal.h"

extern "C" {

namespace {
// the sizes RptSceneOptions has had under this ABI's headers: the first v6 header (before env_park / paths_batch) and today's
constexpr uint32_t OPT_SIZE_V6_FIRST = 104u, OPT_SIZE_NOW = (uint32_t)sizeof(RptSceneOptions);
static_assert(sizeof(RptSceneOptions) == 112, "a grown RptSceneOptions is a new known size: add it to known_opt_size");
bool known_opt_size(uint32_t n) { return n == OPT_SIZE_V6_FIRST || n == OPT_SIZE_NOW; }
void options_default_full(RptSceneOptions* o) {
  std::memset(o, 0, sizeof *o);
  o->struct_size = (uint32_t)sizeof *o;
  o->deep_depth = 8;              // a tree this deep pays for compaction + its own launches
  o->fast_max_depth = (uint32_t)rptdev::KD_MAX_STACK;
  o->sort_rays = -1;
  o->rays_in_kernel = 0;
  o->sort_min_bytes = 8ull << 20;
  o->sort_shadow_min_bytes = 8ull << 20; // (32 MiB until the visibility queries were sized for the shadow rays there are: sorting 40 % fewer keys, the 16k-triangle glass gains from its shadow sort what it lost before — 756 -> 775 Msamples/s)
  o->sort_min_rays = 1u << 19;
  o->nest_trace = 1;
  o->leaf_boxes = 1;
  o->object_filter_min = 5;
  o->device_build_min = 32768;
  o->build_threads = 0;
  o->paths_chunk = 0;
  o->workspace_bytes = 240ull << 30; // (96 GiB until round 6: a 288 GB device ran passes sized for a third of it)
  o->lbuf_bytes = 32ull << 30;
  o->target_paths = 0;
  o->comm_timeout_s = 300.0;
  o->env_park = 1;
}
// the caller's struct may be the smaller one of an older header: never write past ITS size
void copy_options_out(const RptSceneOptions& full, RptSceneOptions* out, uint32_t out_size) {
  std::memcpy(out, &full, out_size);
  out->struct_size = out_size;
}
const char* options_out_of_range(const RptSceneOptions& opt) {
  if (opt.sort_rays < -1 || opt.sort_rays > 1 || opt.deep_depth < 1u || opt.lbuf_bytes < 24u ||
      opt.workspace_bytes < (1ull << 20) || !(opt.comm_timeout_s > 0.0) || (opt.target_paths && opt.target_paths < 1024u) ||
      opt.paths_batch > RPT_PATHS_BATCH_MAX)
    return "RptSceneOptions: a field is out of range";
  return nullptr;
}
} // namespace

void rptgpu_scene_options_default(RptSceneOptions* o) {
  if (!o) return;
  options_default_full(o); // (this header's struct: the full size)
}

int rptgpu_scene_options_default_sized(RptSceneOptions* o, uint32_t struct_size) {
  if (!o) return RPTGPU_E_INVALID_ARGUMENT;
  if (!known_opt_size(struct_size))
    return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "rptgpu_scene_options_default_sized: struct_size is not a size RptSceneOptions has had under this ABI (104, 112)");
  RptSceneOptions full;
  options_default_full(&full);
  copy_options_out(full, o, struct_size);
  return RPTGPU_OK;
}

namespace {
// the environment's overrides of the options (the variables' names: include/rpt_gpu.h, RptSceneOptions), read HERE and
// nowhere else: once per handle, while it is made
void apply_env_overrides(RptSceneOptions& o, bool user_set_build_min) {
  auto ll = [](const char* name, long long& v) { if (const char* e = std::getenv(name)) { v = std::atoll(e); return true; } return false; };
  long long v;
  // several ranks on one node share the host's cores (host_scene.cpp usable_cpus): the device build pays earlier
  for (const char* name : {"RPTGPU_LOCAL_RANKS", "LOCAL_WORLD_SIZE"})
    if (const char* e = std::getenv(name)) {
      if (std::atoi(e) > 1 && !user_set_build_min) o.device_build_min = 4096; // (a default only: a caller's own 32768 stands)
      break;
    }
  if (ll("RPTGPU_DEVICE_BUILD_MIN", v)) o.device_build_min = (uint64_t)std::max(0ll, v);
  if (ll("RPTGPU_BUILD_THREADS", v)) o.build_threads = (uint32_t)std::max(1ll, v);
  if (ll("RPTGPU_FAST_MAX_DEPTH", v)) o.fast_max_depth = (uint32_t)std::max(0ll, v);
  if (ll("RPTGPU_DEEP_DEPTH", v)) o.deep_depth = (uint32_t)std::max(1ll, v);
  if (ll("RPTGPU_RAYS_IN_KERNEL", v)) o.rays_in_kernel = v != 0 ? 1 : 0;
  if (ll("RPTGPU_SORT_RAYS", v)) o.sort_rays = v < 0 ? -1 : (v != 0 ? 1 : 0); // (-1, the documented default: by the tree's footprint)
  if (ll("RPTGPU_SORT_MIN_BYTES", v)) o.sort_min_bytes = (uint64_t)std::max(0ll, v);
  if (ll("RPTGPU_SORT_SHADOW_MIN_BYTES", v)) o.sort_shadow_min_bytes = (uint64_t)std::max(0ll, v);
  if (ll("RPTGPU_SORT_MIN_RAYS", v)) o.sort_min_rays = (uint32_t)std::max(0ll, std::min(v, 0xffffffffll));
  if (ll("RPTGPU_NEST_TRACE", v)) o.nest_trace = v != 0 ? 1 : 0;
  if (ll("RPTGPU_LEAF_BOXES", v)) o.leaf_boxes = v != 0 ? 1 : 0;
  if (ll("RPTGPU_OBJECT_FILTER_MIN", v)) o.object_filter_min = (int32_t)v;
  if (ll("RPTGPU_PATHS_CHUNK", v)) o.paths_chunk = (uint32_t)std::max(0ll, v);
  if (ll("RPTGPU_ENV_PARK", v)) o.env_park = v != 0 ? 1 : 0;
  if (ll("RPTGPU_PATHS_BATCH", v)) o.paths_batch = (uint32_t)std::max(0ll, std::min(v, (long long)RPT_PATHS_BATCH_MAX));
  if (ll("RPTGPU_LBUF_BYTES", v) && v >= 24) o.lbuf_bytes = (uint64_t)v;
  if (const char* e = std::getenv("RPTGPU_TARGET_PATHS")) { uint64_t u = std::strtoull(e, nullptr, 10); if (u >= 1024) o.target_paths = u; }
  if (const char* e = std::getenv("RPTGPU_WS_BYTES")) { uint64_t u = std::strtoull(e, nullptr, 10); if (u >= (1ull << 20)) o.workspace_bytes = u; }
  if (const char* e = std::getenv("RPTGPU_COMM_TIMEOUT_S")) { double d = std::atof(e); if (d > 0.0) o.comm_timeout_s = d; }
}
} // namespace

int rptgpu_scene_get_options(const rptgpu_scene* h, RptSceneOptions* out) {
  if (!h || !out) return RPTGPU_E_INVALID_ARGUMENT;
  // the CALLER says how large its struct is (out->struct_size, set before the call — rptgpu_scene_options_default[_sized]
  // does): a caller built against the 104-byte first v6 header gets 104 bytes, not an overrun of eight
  if (!known_opt_size(out->struct_size))
    return fail(const_cast<rptgpu_scene*>(h), RPTGPU_E_INVALID_ARGUMENT, "rptgpu_scene_get_options: set out->struct_size to sizeof(RptSceneOptions) of your header first (rptgpu_scene_options_default does)");
  copy_options_out(h->opt, out, out->struct_size);
  return RPTGPU_OK;
}

int rptgpu_scene_create(const RptScene* scene, int device, rptgpu_scene** out) {
  return rptgpu_scene_create_opts(scene, device, nullptr, out);
}

int rptgpu_scene_create_opts(const RptScene* scene, int device, const RptSceneOptions* user_opts, rptgpu_scene** out) {
  if (!scene || !out) return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  RptSceneOptions opt;
  options_default_full(&opt);
  bool user_set_build_min = false;
  if (user_opts) { // a caller built against the older (smaller) struct: the fields it does not know keep their defaults
    if (!known_opt_size(user_opts->struct_size)) // (only whole structs: a size in between would cut a field in half)
      return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "RptSceneOptions::struct_size does not belong to this ABI version (use rptgpu_scene_options_default)");
    std::memcpy(&opt, user_opts, user_opts->struct_size);
    opt.struct_size = (uint32_t)sizeof opt;
    if (const char* why = options_out_of_range(opt)) return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, why);
    user_set_build_min = true;
  }
  apply_env_overrides(opt, user_set_build_min);
  if (const char* why = options_out_of_range(opt)) // the overrides are held to the same ranges as the fields
    return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, std::string(why) + " (after the RPTGPU_* environment overrides)");
  opt.fast_max_depth = std::min(opt.fast_max_depth, (uint32_t)rptdev::KD_MAX_STACK);
  rpthost::FlatScene fs;
  std::string err;
  int rc;
  // RPTGPU_PRINT_CREATE=1: where the hand-off's time goes (stderr), for the scene-per-frame use case
  const bool print_create = std::getenv("RPTGPU_PRINT_CREATE") != nullptr;
  auto tc0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!print_create) return;
    auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "scene_create %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tc0).count());
    tc0 = t;
  };
  // Trees of at least RPTGPU_DEVICE_BUILD_MIN primitives (default 32768; 0 = never) are built on the device when there
  // is one — the same tree, an order of magnitude sooner for large meshes (kdbuild.hip); everything else of the
  // flattening, and every validation, is host work.
  rpthost::BuildOptions bopt;
  {
    int nd = 0;
    bopt.device_build_min = (size_t)opt.device_build_min;
    bopt.build_threads = (int)opt.build_threads;
    if (bopt.device_build_min && hipGetDeviceCount(&nd) == hipSuccess && device >= 0 && device < nd) bopt.device = device;
    else (void)hipGetLastError();
  }
  try {
    rc = rpthost::flatten_scene(*scene, fs, err, &bopt); // validates shapes
  } catch (const std::bad_alloc&) {
    return fail(nullptr, RPTGPU_E_OUT_OF_MEMORY, "host allocation failed");
  } catch (...) {
    return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "unexpected exception while flattening");
  }
  if (rc != RPTGPU_OK) return fail(nullptr, rc, err);
  lap(fs.trees_built_on_device ? "flatten + kd build (device)" : "flatten + kd build");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, RPTGPU_E_NO_DEVICE, "no HIP device is visible (hipGetDeviceCount); there is no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "device index out of range");
  rptgpu_scene* h = new (std::nothrow) rptgpu_scene();
  if (!h) return fail(nullptr, RPTGPU_E_OUT_OF_MEMORY, "host allocation failed");
  h->device = device;
  try {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    h->num_cus = prop.multiProcessorCount;
    lap("device, stream, properties");
    h->prefer_wavefront = fs.max_tree_depth >= 3;
    // RPTGPU_FAST_MAX_DEPTH (tests): treat trees deeper than this as too deep for the in-kernel traversals.  The build rule
    // itself keeps real trees far below 32: both children of a median split hold (n + straddlers) / 2 primitives, so a path
    // d levels long needs 16 / 0.85^d primitives with an unsplittable sibling at every level, or 16 * 2^d balanced ones.
    h->opt = opt;
    const uint32_t fast_max_depth = opt.fast_max_depth;
    h->max_tree_depth = fs.max_tree_depth;
    const uint32_t deep_depth = opt.deep_depth;
    h->rays_in_kernel = opt.rays_in_kernel;
    h->sort_mode = opt.sort_rays;
    h->sort_min_bytes = opt.sort_min_bytes;
    h->sort_shadow_min_bytes = opt.sort_shadow_min_bytes;
    h->qtune.sort_min_rays = opt.sort_min_rays;
    h->paths_chunk = opt.paths_chunk;
    h->lbuf_max_bytes = opt.lbuf_bytes;
    h->target_paths = opt.target_paths;
    h->ws_budget_bytes = opt.workspace_bytes;
    for (int i = 0; i < fs.num_objects; i++) {
      const rptdev::Inst& in = fs.insts[i];
      bool tree = in.kind == RPT_SHAPE_MESH || in.kind == RPT_SHAPE_GROUP;
      bool deep = tree && fs.tree_depth[in.tree] >= deep_depth;
      // A group with TREE children (meshes: fractal_teapots.rs; groups: kdtree.rs:14-24 nests without limit) goes through
      // the per-tree kernels whatever its own depth — they are the only ones that walk a tree inside a tree: rpt_nest_trace
      // (two regular levels, one loop) or rpt_tree_generic (anything).  So does a tree deeper than the fast stacks.
      const bool kids = in.kind == RPT_SHAPE_GROUP && fs.tree_kids[in.tree] != 0;
      // deeper than the private stacks of the in-kernel traversals (KD_MAX_STACK): the per-tree kernels, whose stack
      // beyond the LDS levels is a global column as high as the scene's deepest tree (ensure_workspace)
      const bool too_deep = tree && fs.tree_depth[in.tree] > fast_max_depth;
      deep = deep || kids || too_deep;
      h->tree_kids = h->tree_kids || kids || too_deep; // (= some object is for the per-tree pipeline only)
      // rays entering a large tree are sorted by entry cell and octant first: neighbours in a wave then walk the same
      // nodes.  Measured with the VALU-bound traversal kernel of round 2: 100k-triangle mesh (66 MB of nodes + leaf
      // records) 144 -> 172 Msamples/s, 16k-triangle glass (17 MB) 469 -> 528, a 25k-triangle mesh under few bounces
      // (25 MB) 781 -> 766, two 768-triangle meshes (0.6 MB) 4243 -> 3248: the sort sorts EVERY ray of the depth, the
      // gain grows with the work of the rays that enter — so by size, with the threshold well below the glass
      bool sort = false, sort_shadow = false;
      if (deep) {
        const rptdev::Tree& tr = fs.trees[in.tree];
        uint64_t next_node = (size_t)in.tree + 1 < fs.trees.size() ? fs.trees[in.tree + 1].node_base : fs.nodes.size();
        uint64_t next_ref = (size_t)in.tree + 1 < fs.trees.size() ? fs.trees[in.tree + 1].ref_base : fs.refs.size();
        uint64_t bytes = (next_node - tr.node_base) * sizeof(rptdev::KdNode) +
                         (next_ref - tr.ref_base) * (sizeof(uint32_t) + (in.kind == RPT_SHAPE_MESH ? sizeof(rptdev::TriX) : 0));
        sort = h->sort_mode == 1 || (h->sort_mode < 0 && bytes >= h->sort_min_bytes);
        // shadow rays point at ONE light from surfaces that the closest-hit pass just visited in sorted order: for a tree
        // that is not many times the L2s their sort costs more than it gives (16k-triangle glass, ~10 MB: shadow stage
        // 47.7 -> 42.8 ms per two steps without it; 100k-triangle mesh, ~60 MB: 137 -> 180)
        sort_shadow = sort && (h->sort_mode == 1 || bytes >= h->sort_shadow_min_bytes);
      }
      // which traversal kernel of the per-tree pipeline: 1 rpt_tree_trace<TRIS>, 0 rpt_tree_trace over a group of simple
      // shapes, 2 a group with mesh children whose two regular levels fit one traversal stack: rpt_nest_trace
      // (RPTGPU_NEST_TRACE=0: rpt_tree_generic instead), 3 rpt_tree_generic alone (Tree::generic_only)
      uint8_t trace_kind = in.kind == RPT_SHAPE_MESH ? 1 : 0;
      bool generic_only = false;
      if (kids) {
        const rptdev::Tree& tr = fs.trees[in.tree];
        uint32_t inner_depth = 0;
        bool ok = tr.regular && !(fs.tree_kids[in.tree] & 2u); // rpt_nest_trace: no group children, no irregular trees
        for (uint32_t k = 0; k < tr.num_prims; k++) {
          const rptdev::Inst& kid = fs.insts[tr.prim_base + k];
          if (kid.kind == RPT_SHAPE_MESH) {
            inner_depth = std::max(inner_depth, fs.tree_depth[kid.tree]);
            ok = ok && fs.trees[kid.tree].regular;
          }
        }
        if (ok && opt.nest_trace != 0 && fs.tree_depth[in.tree] + inner_depth + 2 <= (uint32_t)rptdev::KD_MAX_STACK) trace_kind = 2;
        else generic_only = true;
      }
      if (generic_only) {
        trace_kind = 3;
        fs.trees[in.tree].generic_only = 1u;
        sort = false;
        sort_shadow = false;
      }
      h->sort_rays = h->sort_rays || sort;
      // obj_deep: 0 in-kernel; 1 per-tree; 2 per-tree with the ray sort; +4: every ray of it goes through rpt_tree_generic
      // (an irregular tree, an object only that kernel is built for)
      const bool all_generic = deep && (generic_only || !fs.trees[in.tree].regular);
      h->gen_all = h->gen_all || all_generic;
      // +8: the sort serves the closest-hit query only
      h->obj_deep.push_back(deep ? (uint8_t)((sort ? 2 : 1) | (all_generic ? 4 : 0) | (sort && !sort_shadow ? 8 : 0)) : 0);
      // bit 4 (shallow objects): a primitive or a tree that is ONE leaf — runs of such objects take the lean build of
      // rpt_rays_objects (kernels/wavefront.inc)
      const bool one_leaf = !tree || fs.trees[in.tree].root_leaf != 0;
      h->obj_tris.push_back((uint8_t)(trace_kind | (!deep && one_leaf ? 16 : 0)));
      h->has_deep = h->has_deep || deep;
    }
    h->gen_levels = fs.generic_levels; h->gen_frames = fs.generic_frames;
    // scenes whose trees are all walked inside rpt_extend / rpt_shadow_rays get their paths re-ordered per depth
    // (RPTGPU_PATH_REORDER: an A/B switch, environment only — scheduling, not results)
    std::memcpy(h->scene_bounds, fs.scene_bounds, sizeof h->scene_bounds);
    h->path_reorder = !h->has_deep && fs.scene_bounds_ok && !fs.trees.empty();
    if (const char* e = std::getenv("RPTGPU_PATH_REORDER")) h->path_reorder = h->path_reorder && std::atoi(e) != 0;
    if (const char* e = std::getenv("RPTGPU_PATH_REORDER_MIN")) h->path_reorder_min = (uint32_t)std::max(1, std::atoi(e));
    if (h->tree_kids) h->prefer_wavefront = true;
    h->all_flat = true;
    for (const rptdev::Tree& tr : fs.trees) h->all_flat = h->all_flat && tr.root_leaf != 0;
    if (h->all_flat) h->path_reorder = false; // every tree a single leaf: nothing in rpt_extend diverges by where a ray goes
    if (h->all_flat) { // does the scene fit a wave's share of LDS (160 KB per CU / 8 waves)?
      constexpr uint32_t WAVE_LDS = RPT_PATHS_WAVE_LDS - RPT_PATHS_WALKER_LDS; // the wave's share less the fold walker's state
      auto up16 = [](uint64_t v) { return (v + 15) & ~15ull; };
      uint64_t off = 0;
      FlatLayout lay{};
      lay.n_refs = (uint32_t)fs.refs.size();
      // intersection records, leaf entries and materials are what a query reads; the triangles themselves (vertex
      // normals of the hit that stands, light sampling) join them only if everything still fits — C2 does (12
      // triangles), a room of 23 polygons keeps them in global memory and is flat all the same
      auto assign = [&](bool with_tris) {
        lay.n_tris = with_tris ? (uint32_t)fs.tris.size() : 0u;
        off = up16(fs.refs.size() * sizeof(rptdev::TriX));
        lay.off_tris = (uint32_t)off; off = up16(off + (uint64_t)lay.n_tris * sizeof(rptdev::Tri));
        lay.off_refs = (uint32_t)off; off = up16(off + fs.refs.size() * sizeof(uint32_t));
        lay.off_mat = (uint32_t)off;  off = up16(off + (uint64_t)fs.num_objects * sizeof(rptdev::Material));
        lay.off_leaf = (uint32_t)off; off = up16(off + (uint64_t)fs.num_objects * 16);
      };
      assign(true); // (rpt_paths<KdFlat>: that instantiation also stashes camera rays in LDS)
      if (off + 12 * 64 * sizeof(double) + RPT_PATHS_STASH_LDS > WAVE_LDS || std::getenv("RPTGPU_FLAT_TRIS_GLOBAL")) assign(false); // (room for the plane table)
      // shared slab quotients: distinct plane coordinates per axis over the untransformed meshes (bitwise
      // distinct: -0.0 and 0.0 give differently signed zeros), at most 4 per axis or the feature stays off
      std::vector<double> planes(12, 0.0);
      uint32_t cnt[3] = {0, 0, 0};
      bool planes_ok = true;
      auto slot_of = [&](int axis, double v) -> int {
        uint64_t bits;
        std::memcpy(&bits, &v, 8);
        for (uint32_t j = 0; j < cnt[axis]; j++) {
          uint64_t b2;
          std::memcpy(&b2, &planes[axis * 4 + j], 8);
          if (b2 == bits) return axis * 4 + (int)j;
        }
        if (cnt[axis] == 4) return -1;
        planes[axis * 4 + cnt[axis]] = v;
        return axis * 4 + (int)cnt[axis]++;
      };
      static const int FACE[6] = {0, 3, 1, 4, 2, 5}; // bounds[] index of the faces in div6's order
      std::vector<uint32_t> idx(fs.num_objects, 0);
      for (int i = 0; i < fs.num_objects && planes_ok; i++) {
        const rptdev::Inst& in = fs.insts[i];
        if (in.kind != RPT_SHAPE_MESH || in.has_xf) continue;
        for (int k = 0; k < 6; k++) {
          int sl = slot_of(FACE[k] % 3, in.bounds[FACE[k]]);
          if (sl < 0) { planes_ok = false; break; }
          idx[i] |= (uint32_t)sl << (4 * k);
        }
      }
      if (planes_ok && cnt[0] + cnt[1] + cnt[2] > 0 && !std::getenv("RPTGPU_NO_PLANE_TABLE")) {
        for (int i = 0; i < fs.num_objects; i++) {
          rptdev::Inst& in = fs.insts[i];
          if (in.kind == RPT_SHAPE_MESH && !in.has_xf) { in.plane_idx = idx[i]; in.plane_use = 1; }
        }
        // plane_use = number of consecutive table users starting here, capped at the device's run length
        for (int i = fs.num_objects - 1; i >= 0; i--) {
          rptdev::Inst& in = fs.insts[i];
          if (!in.plane_use) continue;
          uint32_t next = (i + 1 < fs.num_objects) ? fs.insts[i + 1].plane_use : 0u;
          in.plane_use = std::min<uint32_t>((uint32_t)RPT_FLAT_RUN, 1u + next);
        }
        lay.plane_cnt = cnt[0] | (cnt[1] << 4) | (cnt[2] << 8);
        // the table's slots are packed (x planes, then y, then z): plane_idx goes from axis * 4 + j to that numbering
        const uint32_t base[3] = {0u, cnt[0], cnt[0] + cnt[1]};
        for (int i = 0; i < fs.num_objects; i++) {
          rptdev::Inst& in = fs.insts[i];
          if (!in.plane_use) continue;
          uint32_t packed = 0;
          for (int k = 0; k < 6; k++) {
            const uint32_t sl = (in.plane_idx >> (4 * k)) & 15u;
            packed |= (base[sl >> 2] + (sl & 3u)) << (4 * k);
          }
          in.plane_idx = packed;
        }
        lay.off_qtab = (uint32_t)off; off = up16(off + (uint64_t)(cnt[0] + cnt[1] + cnt[2]) * 64 * sizeof(double));
        h->plane_vals.upload(planes, h->stream);
        HIP_TRY(hipStreamSynchronize(h->stream)); // `planes` dies with this block
        lay.plane_vals = h->plane_vals.p;
      }
      // many small objects and no plane table (a room of polygons rather than C2's five walls): the object filter
      // (host_scene.cpp fill_object_boxes).  RPTGPU_OBJECT_FILTER_MIN: from how many objects (0 = never).  Measured:
      // 2 objects -5..-11 % (C1, glass spheres), 5 objects +8 % (basic.rs), 6 objects +4 % (spheres.rs), 29 objects +40 %
      {
        const int min_objects = opt.object_filter_min;
        const uint64_t every = fs.num_objects >= 64 ? ~0ull : (1ull << fs.num_objects) - 1ull;
        if (!lay.plane_cnt && min_objects > 0 && fs.num_objects >= min_objects && fs.obj_filter_ok &&
            (fs.obj_always & every) != every) {
          // rpt_paths<KdFlatF> reads triangles from global memory (no plane table here, so `off` is final)
          const FlatLayout keep = lay;
          const uint64_t keep_off = off;
          if (lay.n_tris) assign(false);
          const uint64_t with_boxes = up16(off + (uint64_t)fs.num_objects * 6 * sizeof(double));
          if (with_boxes <= WAVE_LDS) {
            lay.obj_filter = 1;
            lay.obj_always = fs.obj_always & every;
            lay.off_obox = (uint32_t)off; off = with_boxes;
            h->obj_box.upload(fs.obj_lbox, h->stream);
            std::vector<double> grid(fs.obj_grid, fs.obj_grid + 12);
            h->obj_grid.upload(grid, h->stream);
            HIP_TRY(hipStreamSynchronize(h->stream)); // `grid` dies with this block
            lay.obj_box = h->obj_box.p;
            lay.obj_grid = h->obj_grid.p;
          } else {
            lay = keep;
            off = keep_off;
          }
        }
      }
      lay.off_end = (uint32_t)off;
      if (off > WAVE_LDS) {
        h->all_flat = false;
      } else {
        h->flat_layout = lay;
      }
    }
    lap("pipeline choice, flat layout");
    h->ext_shapes = fs.nested_mesh;
    for (const rptdev::Inst& in : fs.insts) h->ext_shapes = h->ext_shapes || in.kind == RPT_SHAPE_MONOMIAL;
    for (const rptdev::Light& l : fs.lights) h->light_casts.push_back(l.kind != RPT_LIGHT_AMBIENT ? 1 : 0);
    h->host_lights = fs.lights;
    h->insts.upload(fs.insts, h->stream);
    h->trees.upload(fs.trees, h->stream);
    h->nodes.upload(fs.nodes, h->stream);
    h->refs.upload(fs.refs, h->stream);
    h->tris.upload(fs.tris, h->stream);
    h->trix.upload(fs.lrec, h->stream);
    h->lbox.upload(fs.lbox, h->stream);
    h->materials.upload(fs.materials, h->stream);
    h->lights.upload(fs.lights, h->stream);
    h->env_texels.upload(fs.env_texels, h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
    lap("device allocation + upload");
    rptdev::Scene& d = h->dscene;
    d.insts = h->insts.p; d.trees = h->trees.p; d.nodes = h->nodes.p; d.refs = h->refs.p; d.tris = h->tris.p; d.lrec = h->trix.p; d.lbox = h->lbox.p;
    d.materials = h->materials.p; d.lights = h->lights.p; d.env_texels = h->env_texels.p;
    std::memcpy(d.env_color, fs.env_color, sizeof d.env_color);
    d.env_width = fs.env_width; d.env_height = fs.env_height; d.env_kind = fs.env_kind;
    d.num_objects = fs.num_objects; d.num_lights = (int32_t)fs.lights.size();
    d.num_shadow_lights = fs.num_shadow_lights;
    d.use_leaf_boxes = opt.leaf_boxes != 0 ? 1 : 0;
  } catch (const HipError& e) {
    int code = hip_fail(nullptr, e);
    delete h;
    return code;
  }
  *out = h;
  return RPTGPU_OK;
}

void rptgpu_scene_destroy(rptgpu_scene* h) { delete h; }

// KdBuild -> the malloc'ed arrays of RptKdTree
static int kdtree_export(const rpthost::KdBuild& kb, RptKdTree* out) {
  size_t nn = kb.nodes.size(), nr = kb.refs.size();
  out->split = (double*)std::malloc(std::max<size_t>(nn, 1) * sizeof(double));
  out->info = (uint32_t*)std::malloc(std::max<size_t>(nn, 1) * sizeof(uint32_t));
  out->a = (uint32_t*)std::malloc(std::max<size_t>(nn, 1) * sizeof(uint32_t));
  out->b = (uint32_t*)std::malloc(std::max<size_t>(nn, 1) * sizeof(uint32_t));
  out->refs = (uint32_t*)std::malloc(std::max<size_t>(nr, 1) * sizeof(uint32_t));
  if (!out->split || !out->info || !out->a || !out->b || !out->refs) {
    rptgpu_kdtree_free(out);
    return RPTGPU_E_OUT_OF_MEMORY;
  }
  for (size_t i = 0; i < nn; i++) {
    out->split[i] = kb.nodes[i].split;
    out->info[i] = kb.nodes[i].ib & 3u;
    out->a[i] = kb.nodes[i].a;
    out->b[i] = kb.nodes[i].ib >> 2;
  }
  std::memcpy(out->refs, kb.refs.data(), nr * sizeof(uint32_t));
  out->num_nodes = nn;
  out->num_refs = nr;
  out->max_depth = kb.max_depth;
  out->regular = kb.regular ? 1u : 0u;
  return RPTGPU_OK;
}

int rptgpu_kdtree_build(const double* boxes, uint64_t n, RptKdTree* out) {
  if (!out || (n && !boxes)) return RPTGPU_E_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  try {
    std::vector<rpthost::Box> b(n);
    for (uint64_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        b[i].lo[k] = boxes[6 * i + k];
        b[i].hi[k] = boxes[6 * i + 3 + k];
      }
    rpthost::KdBuild kb;
    rpthost::kd_build(b, kb);
    return kdtree_export(kb, out);
  } catch (...) {
    return RPTGPU_E_OUT_OF_MEMORY;
  }
}

int rptgpu_kdtree_build_device(const double* boxes, uint64_t n, int device, RptKdTree* out) {
  if (!out || (n && !boxes)) return RPTGPU_E_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
    return fail(nullptr, RPTGPU_E_NO_DEVICE, "no HIP device is visible (hipGetDeviceCount); there is no CPU fallback");
  if (device < 0 || device >= nd) return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "device index out of range");
  try {
    std::vector<rpthost::Box> b(n);
    for (uint64_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        b[i].lo[k] = boxes[6 * i + k];
        b[i].hi[k] = boxes[6 * i + 3 + k];
      }
    rpthost::KdBuild kb;
    std::string why;
    if (!rpthost::kd_build_device(b, kb, device, why))
      return fail(nullptr, RPTGPU_E_INVALID_ARGUMENT, "the device kd build does not take this input: " + why);
    return kdtree_export(kb, out);
  } catch (...) {
    return RPTGPU_E_OUT_OF_MEMORY;
  }
}

void rptgpu_kdtree_free(RptKdTree* t) {
  if (!t) return;
  std::free(t->split); std::free(t->info); std::free(t->a); std::free(t->b); std::free(t->refs);
  std::memset(t, 0, sizeof *t);
}

} // extern "C"
