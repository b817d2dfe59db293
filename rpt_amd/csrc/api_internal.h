// api_internal.h — what the files behind the C ABI of include/rpt_gpu.h share: the handle, device buffers, the error
// convention, the RCCL binding.  The ABI's implementation is split by concern (round 6; it was one 1 900-line api.cpp):
//   api_common.cpp   errors, the RCCL loader, kernel tables, version / strerror / stats
//   api_scene.cpp    RptSceneOptions, rptgpu_scene_create[_opts] (flattening, routing, upload), the kd-tree entry points
//   api_render.cpp   workspace, the wavefront loop and the persistent launch (render_impl), render_batch[_device],
//                    rptgpu_closest_hit, rptgpu_eval_math
//   api_comm.cpp     communicator, rptgpu_render_batch_reduce (the library-owned exchange and its failure paths),
//                    rptgpu_render_batch_emulate_ranks
//   api_buffer.cpp   the device-resident Buffer
// No compute happens on the host; if there is no HIP device every compute entry point returns RPTGPU_E_NO_DEVICE (there is
// no CPU fallback by design).
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rpt_gpu.h"
#include "device_types.h"
#include "host_scene.h"
#include "kernels.h"

namespace rptapi {

extern thread_local std::string g_create_error;

struct HipError {
  hipError_t e;
  const char* what;
  int line;
  const char* file = "";
};
// the file's name without its directories, for error messages
constexpr const char* rpt_basename(const char* path) {
  const char* b = path;
  for (const char* p = path; *p; p++)
    if (*p == '/') b = p + 1;
  return b;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) throw HipError{_e, #expr, __LINE__, rptapi::rpt_basename(__FILE__)}; \
  } while (0)

// owning device allocation: released by the destructor, never copied
template <class T> struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    if (count <= n && p) return;
    release();
    HIP_TRY(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
    n = count;
  }
  void upload(const std::vector<T>& v, hipStream_t st) {
    alloc(v.size());
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

constexpr int MAX_EVENT_PAIRS = 4096;
// ---- RCCL, opened on first use (the library has no link-time dependency on it) -----------------------------
// the handful of declarations of <rccl/rccl.h> that are used here
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*CommAbort)(RcclComm) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, int, RcclComm, hipStream_t) = nullptr;
  // the gather (optional: without them rptgpu_render_batch_reduce falls back to the reduce)
  int (*Send)(const void*, size_t, int, int /*peer*/, RcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int /*peer*/, RcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommGetAsyncError)(RcclComm, int*) = nullptr; // optional: polled while a batch's collective is in flight
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0, RCCL_IN_PROGRESS = 7; // ncclFloat32, ncclSum, ncclInProgress (rccl.h)
Rccl& rccl(); // opened on first use (api_common.cpp)

} // namespace rptapi
using namespace rptapi;

struct rptgpu_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  // flattened scene on the device
  DevBuf<rptdev::Inst> insts;
  DevBuf<rptdev::Tree> trees;
  DevBuf<rptdev::KdNode> nodes;
  DevBuf<uint32_t> refs;
  DevBuf<rptdev::Tri> tris;
  DevBuf<rptdev::TriX> trix;
  DevBuf<rptdev::LeafBox> lbox;
  DevBuf<rptdev::LeafBox> obj_box; // flat scenes' object filter (FlatLayout::obj_box / obj_grid)
  DevBuf<double> obj_grid;
  DevBuf<rptdev::Material> materials;
  DevBuf<rptdev::Light> lights;
  DevBuf<double> env_texels;
  rptdev::Scene dscene{};
  // workspace
  DevBuf<double> ray, hit, rec, shadow, accum, out_full;
  DevBuf<int32_t> hit_obj;
  DevBuf<uint32_t> draw, counters, pixels;
  uint64_t ws_cap = 0;
  uint64_t ws_rec_cols = 0;            // columns of the depth-record pool (PathState::rec)
  DevBuf<uint32_t> rec_parent, last_col;
  DevBuf<double> ray_next;             // dense path state of the NEXT depth (PathState: rpt_shade writes, the host swaps)
  DevBuf<uint32_t> draw_next, pid, pid_next, col, col_next;
  // in-kernel-traversal scenes: the paths of a depth are re-ordered by ray key (kernels/wavefront.inc rpt_path_keys)
  bool path_reorder = false;
  uint32_t path_reorder_min = RPT_PATH_REORDER_MIN; // ... when a depth has at least this many (RPTGPU_PATH_REORDER_MIN: tests)
  double scene_bounds[6] = {0, 0, 0, 1, 1, 1};
  DevBuf<uint32_t> path_order;
  DevBuf<double> next_rows;           // [cap][8]: the survivors' next state as rows (PathState::next_rows)
  double rec_ratio = 0.0;              // record columns a path of this scene needs on average, as measured by the passes so
  uint32_t rec_ratio_bounces = 0xffffffffu; // far at this max_bounces (0 = not measured yet: the next pass measures)
  uint64_t ws_fail_paths = 0;          // the smallest pass (paths) whose workspace did not fit on this device so far; 0 = none
  DevBuf<double> prec;                 // persistent kernel: depth records [threads][bounces][8]
  DevBuf<double> lbuf;                 // persistent kernel: radiance of every sample of a launch [spp][3][npix]
  uint64_t lbuf_max_bytes = 32ull << 30; // cap on lbuf (RPTGPU_LBUF_BYTES); larger batches run as several launches
  uint32_t paths_chunk = 0;            // samples per work item (RptSceneOptions::paths_chunk; 0 = chosen per launch)
  DevBuf<unsigned long long> pcounters; // [0] closest-hit rays [1] shadow rays
  int num_cus = 0;
  bool prefer_wavefront = false; // scene has real kd-trees: traversal-latency bound
  bool all_flat = false;         // every tree is a single leaf (and the scene fits the LDS tables): the path kernel
                                 // without any traversal code
  FlatLayout flat_layout{};        // the flat kernel's dynamic LDS
  DevBuf<double> plane_vals;       // distinct bounding-plane coordinates of the untransformed meshes [3][4]
  uint32_t flat_lds_bytes = 0;
  bool ext_shapes = false;       // scene has a shape only the *_ext kernel builds implement
  // deep-tree scenes: per top-level object flags and the buffers of the object-by-object query
  std::vector<uint8_t> obj_deep, obj_tris, light_casts;
  // every per-tree object of the scene is one only by the kd-trees-of-kd-trees rule (shallow group, mesh children): for
  // (fractal_teapots at 8 bounces, 3.8 M paths per pass: 60.7 against 56.9 Msamples/s; 7.7 M: 70.6 against 86.7)
  bool tree_kids = false;   // some object is a group with tree children, or a tree too deep for the in-kernel
                            // traversals: only the per-tree pipeline walks those
  uint32_t max_tree_depth = 0;
  uint32_t gen_levels = 0, gen_frames = 0; // rpt_tree_generic's column heights for this scene (host_scene.cpp)
  bool gen_all = false;     // some object sends EVERY ray through rpt_tree_generic (irregular tree, generic_only)
  DevBuf<double> gen_defer, gen_frame;
  DevBuf<uint32_t> gen_overflow;
  uint32_t gen_threads = 0;
  std::vector<rptdev::Light> host_lights; // (what launch decisions need of the lights)
  std::vector<uint32_t> cnt_host;  // the per-depth counters read back from the device
  bool has_deep = false;
  int rays_in_kernel = 0;          // RPTGPU_RAYS_IN_KERNEL: rptgpu_closest_hit keeps to rpt_extend_rays also when the scene has deep trees
  DevBuf<uint32_t> tq, tq_ctr;
  StackSpill spill{};              // the per-tree traversal kernels' stack beyond the LDS levels (kernels.h)
  DevBuf<uint32_t> spill_node;
  DevBuf<double> spill_ts, spill_bmax;
  DevBuf<double> tree_rays;        // [cap][8]: rpt_tree_enter's rows for the traversal kernels' refill
  // optional ray sort in front of the per-tree traversal (RPTGPU_SORT_RAYS)
  bool sort_rays = false;          // some deep tree is large enough for sorting to pay (obj_deep[i] == 2)
  int sort_mode = -1;              // RPTGPU_SORT_RAYS: 0 never, 1 every deep tree, default: by footprint
  uint64_t sort_min_bytes = 8ull << 20;  // RPTGPU_SORT_MIN_BYTES: nodes + leaf records of a tree whose rays are worth sorting
  RptSceneOptions opt{};           // the handle's knobs: defaults, the caller's RptSceneOptions, environment overrides — fixed at creation
  QueryTuning qtune{0u, 1u << 19};  // launch_query's counter-set toggle; opt.sort_min_rays
  uint64_t sort_shadow_min_bytes = 8ull << 20; // RPTGPU_SORT_SHADOW_MIN_BYTES: ... whose SHADOW rays are, too
  DevBuf<uint32_t> sort_kin, sort_kout, sort_vin;
  DevBuf<uint8_t> sort_tmp;
  SortBufs sort_bufs{};
  DevBuf<double> srt;
  DevBuf<uint32_t> shadow_q;
  // cached pixel partition
  uint32_t part_key[6] = {0, 0, 0, 0, 0, 0};
  uint32_t npix = 0;
  // accounting
  RptStats stats{};
  std::vector<hipEvent_t> ev_pool;
  struct Pending { int kind; int e0, e1; };
  std::vector<Pending> pending;
  int ev_used = 0;
  uint64_t target_paths = 0;       // RPTGPU_TARGET_PATHS: paths in flight per pass of the wavefront pipeline; 0 = as many as
                                   // the workspace budget holds (ws_budget_bytes and half of the free HBM), at most 128 Mi
  uint64_t ws_budget_bytes = 240ull << 30; // RPTGPU_WS_BYTES
  // multi-GPU: the communicator of this handle (rptgpu_comm_init) and its frame buffers
  RcclComm comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  DevBuf<float> frame32, frame32_sum;
  // multi-GPU gather: this rank's packed pixels; on the root the peers' packed pixels and every rank's pixel list
  DevBuf<float> packed32, gather32;
  DevBuf<uint32_t> gather_pixels;          // [world] lists back to back, in rank order
  std::vector<uint64_t> gather_off;        // [world + 1] offsets (pixels) into gather_pixels
  uint32_t gather_key[5] = {0, 0, 0, 0, 0}; // width, height, world, root, 1
  bool comm_failed = false;                // a batch's collective failed: sticky until comm_destroy + comm_init
  bool abandoned = false;                  // an aborted batch's work did not drain: kernels of it may still run on the old stream
                                           // and touch the workspace — nothing more is enqueued on this handle, ever
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};

  ~rptgpu_scene() {
    (void)hipSetDevice(device);
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
    if (comm && rccl().ok) (void)rccl().CommDestroy(comm);
    if (stream) (void)hipStreamDestroy(stream);
    // every DevBuf member frees itself (its destructor runs after this body, on `device`)
  }
};

namespace rptapi {

// the error convention: every entry point returns an int; the detail goes to the handle (or, without one, to the thread)
int fail(rptgpu_scene* h, int code, const std::string& detail);
int hip_fail(rptgpu_scene* h, const HipError& e);
// ext: the scene contains a shape of the extended set (RPT_SHAPE_MONOMIAL), which only the *_ext builds
// of the kernels know; everything else runs the base builds
const KernelTable* table_for(uint32_t mode, bool ext = false);
extern const char* const BAD_MODE;

// api_render.cpp
void ensure_partition(rptgpu_scene* h, const RptRenderParams& p);
std::vector<uint32_t> pixel_list(uint32_t width, uint32_t height, uint32_t tw, uint32_t th, uint32_t pi, uint32_t pc);
const char* bad_params(const RptRenderParams* p);
// packed (with d_out, f32 or f64): d_out receives only this part's pixels, [npix][3] in the order of the part's pixel list
int render_impl(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* p, void* d_out, bool out_f32,
                double* host_out, hipStream_t user_stream, bool packed = false);

// A handle whose aborted batch never drained (rptgpu_render_batch_reduce, drain_after_abort): the abandoned stream's
// kernels may still read and write the workspace, the frame buffers and events, so every call that would enqueue work
// refuses — until rptgpu_scene_destroy.
#define REFUSE_IF_ABANDONED(h)                                                                                            \
  do {                                                                                                                    \
    if ((h) && (h)->abandoned)                                                                                            \
      return fail((h), RPTGPU_E_COMM, "an aborted batch's device work never drained on this handle: destroy it (its "    \
                                      "workspace may still be written by the abandoned stream)");                        \
  } while (0)


} // namespace rptapi
