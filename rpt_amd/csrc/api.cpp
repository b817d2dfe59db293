// api.cpp — the C ABI of include/rpt_gpu.h: device memory, the wavefront loop that drives the
// gfx950 kernels, accounting.  No compute happens on the host; if there is no HIP device every
// compute entry point returns RPTGPU_E_NO_DEVICE (there is no CPU fallback by design).
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

namespace {

thread_local std::string g_create_error;

struct HipError {
  hipError_t e;
  const char* what;
  int line;
};
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) throw HipError{_e, #expr, __LINE__}; \
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
Rccl load_rccl() {
  Rccl r;
  // RPTGPU_FAIL_COMM=1: test hook — behave as if librccl.so could not be opened, so that the error paths of
  // rptgpu_comm_unique_id / rptgpu_comm_init (and bench.py's fallback) can be exercised on any box
  if (const char* e = std::getenv("RPTGPU_FAIL_COMM"); e && std::atoi(e) != 0) {
    r.why = "RPTGPU_FAIL_COMM is set (test hook): RCCL treated as unavailable";
    return r;
  }
  std::string err;
  // A copy the process already has (PyTorch-ROCm brings its own librccl.so) is used as it is; otherwise ours is opened
  // RTLD_LOCAL: a second librccl.so loaded later by someone else must not bind its symbols to this one — two copies
  // with RTLD_GLOBAL ended in "double free or corruption" at process exit (pytest importing torch after the first
  // rptgpu_comm_* call)
  for (const char* name : {"librccl.so", "librccl.so.1"}) {
    r.so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (r.so) break;
  }
  for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    if (r.so) break;
    r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.so) break;
    const char* e = dlerror(); // ONE call: dlerror() clears the message it returns
    if (e && err.empty()) err = e;
  }
  if (!r.so) { r.why = "dlopen(librccl.so): " + (err.empty() ? std::string("not found") : err); return r; }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
  r.CommAbort = (decltype(r.CommAbort))dlsym(r.so, "ncclCommAbort"); // optional: the failure path of the reduce
  r.Reduce = (decltype(r.Reduce))dlsym(r.so, "ncclReduce");
  r.Send = (decltype(r.Send))dlsym(r.so, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(r.so, "ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.so, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.so, "ncclGroupEnd");
  r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(r.so, "ncclCommGetAsyncError");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Reduce;
  if (!r.ok) r.why = "librccl.so lacks an expected symbol";
  return r;
}
Rccl& rccl() {
  static Rccl r = load_rccl(); // function-local static: initialised once, thread-safe (C++11)
  return r;
}

} // namespace

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
  DevBuf<uint32_t> draw, queue_a, queue_b, counters, pixels;
  DevBuf<uint8_t> nrec;
  uint64_t ws_cap = 0;
  uint32_t ws_bounces = 0;
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
  uint64_t ws_budget_bytes = 96ull << 30; // RPTGPU_WS_BYTES
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

namespace {

int fail(rptgpu_scene* h, int code, const std::string& detail) {
  if (h) h->error = detail;
  else g_create_error = detail;
  return code;
}

int hip_fail(rptgpu_scene* h, const HipError& e) {
  char buf[512];
  std::snprintf(buf, sizeof buf, "%s failed at api.cpp:%d: %s", e.what, e.line, hipGetErrorString(e.e));
  int code = (e.e == hipErrorOutOfMemory) ? RPTGPU_E_OUT_OF_MEMORY
             : (e.e == hipErrorNoDevice || e.e == hipErrorInvalidDevice || e.e == hipErrorInsufficientDriver)
                 ? RPTGPU_E_NO_DEVICE
                 : RPTGPU_E_HIP;
  return fail(h, code, buf);
}

// ext: the scene contains a shape of the extended set (RPT_SHAPE_MONOMIAL), which only the *_ext builds
// of the kernels know; everything else runs the base builds
const KernelTable* table_for(uint32_t /*mode: RPT_PRECISION_F64_STRICT is the only one*/, bool ext = false) {
  return ext ? &rpt_strict_ext::TABLE : &rpt_strict::TABLE;
}
const char* const BAD_MODE = "unknown precision_mode (RPT_PRECISION_F64_STRICT = 0 is the only mode; F64_FAST was removed in ABI v4)";

// profiling: bracket a launch with two events from a pool; resolved at the end of the call
struct Bracket {
  rptgpu_scene* h;
  int kind;
  bool on;
  int e0 = -1;
  Bracket(rptgpu_scene* h_, int kind_, bool on_) : h(h_), kind(kind_), on(on_) {
    h->stats.kernel_launches[kind]++;
    if (!on) return;
    if (h->ev_used + 2 > MAX_EVENT_PAIRS * 2) { on = false; return; }
    while ((int)h->ev_pool.size() < h->ev_used + 2) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      h->ev_pool.push_back(e);
    }
    e0 = h->ev_used;
    h->ev_used += 2;
    HIP_TRY(hipEventRecord(h->ev_pool[e0], h->stream));
  }
  void done() {
    if (!on) return;
    HIP_TRY(hipEventRecord(h->ev_pool[e0 + 1], h->stream));
    h->pending.push_back({kind, e0, e0 + 1});
  }
};

// launch_query's accounting hook: phases of a query bracketed with pool events like every other launch
struct QueryMarks {
  rptgpu_scene* h;
  bool on;
  int e0[RPT_K_COUNT];
  QueryMarks(rptgpu_scene* h_, bool on_) : h(h_), on(on_) {
    for (int& e : e0) e = -1;
  }
};
void query_mark(void* ctx, int kind, int end) {
  QueryMarks* q = (QueryMarks*)ctx;
  if (kind < 0 || kind >= RPT_K_COUNT) return;
  rptgpu_scene* h = q->h;
  if (!end) {
    h->stats.kernel_launches[kind]++;
    q->e0[kind] = -1;
    if (!q->on || h->ev_used + 2 > MAX_EVENT_PAIRS * 2) return;
    while ((int)h->ev_pool.size() < h->ev_used + 2) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      h->ev_pool.push_back(e);
    }
    q->e0[kind] = h->ev_used;
    h->ev_used += 2;
    HIP_TRY(hipEventRecord(h->ev_pool[q->e0[kind]], h->stream));
  } else if (q->e0[kind] >= 0) {
    HIP_TRY(hipEventRecord(h->ev_pool[q->e0[kind] + 1], h->stream));
    h->pending.push_back({kind, q->e0[kind], q->e0[kind] + 1});
    q->e0[kind] = -1;
  }
}

void drain_events(rptgpu_scene* h) {
  for (auto& p : h->pending) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[p.e0], h->ev_pool[p.e1]));
    h->stats.kernel_ms[p.kind] += ms;
  }
  h->pending.clear();
  h->ev_used = 0;
}

// the pixels of part pi of pc, in the order the path kernels walk them: 8x8-pixel blocks, row-major inside a block —
// the 64 lanes of a wave start on one compact block, so their paths see the same part of the scene (coherent
// traversal, similar lengths)
std::vector<uint32_t> pixel_list(uint32_t width, uint32_t height, uint32_t tw, uint32_t th, uint32_t pi, uint32_t pc) {
  std::vector<uint32_t> pix;
  uint32_t tiles_x = (width + tw - 1) / tw;
  pix.reserve((size_t)width * height / pc + 1);
  for (uint32_t by = 0; by < height; by += 8)
    for (uint32_t bx = 0; bx < width; bx += 8)
      for (uint32_t y = by; y < std::min(by + 8, height); y++)
        for (uint32_t x = bx; x < std::min(bx + 8, width); x++) {
          uint32_t tile = (y / th) * tiles_x + (x / tw);
          if (pc <= 1 || tile % pc == pi) pix.push_back(y * width + x);
        }
  return pix;
}
void ensure_partition(rptgpu_scene* h, const RptRenderParams& p) {
  uint32_t tw = p.tile_width ? p.tile_width : 32, th = p.tile_height ? p.tile_height : 8;
  uint32_t pc = p.part_count ? p.part_count : 1, pi = p.part_count ? p.part_index : 0;
  uint32_t key[6] = {p.width, p.height, tw, th, pi, pc};
  if (std::memcmp(key, h->part_key, sizeof key) == 0 && h->pixels.p) return;
  std::vector<uint32_t> pix = pixel_list(p.width, p.height, tw, th, pi, pc);
  h->pixels.upload(pix, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->npix = (uint32_t)pix.size();
  std::memcpy(h->part_key, key, sizeof key);
}

// rpt_tree_generic's columns: a small grid for the few rays the fast kernels hand on, a large one when whole objects
// (or, under RPT_FLAG_GENERAL_TRAVERSAL, everything) go through it.  Heights: what the scene's deepest nest needs.
void ensure_generic(rptgpu_scene* h, bool all) {
  const uint32_t blocks_few = 64, blocks_all = (uint32_t)std::max(64, std::min(1024, h->num_cus * 4));
  const uint32_t want = (all ? blocks_all : blocks_few) * 256u;
  if (h->gen_threads < want) {
    const uint64_t levels = std::max(1u, h->gen_levels), frames = std::max(1u, h->gen_frames);
    h->gen_defer.release(); h->gen_frame.release();
    h->gen_defer.alloc(levels * 8u * want);
    h->gen_frame.alloc(frames * 12u * want);
    h->gen_overflow.alloc(1);
    HIP_TRY(hipMemsetAsync(h->gen_overflow.p, 0, sizeof(uint32_t), h->stream));
    h->gen_threads = want;
  }
  h->spill.gen = GenericStack{h->gen_defer.p, h->gen_frame.p, h->gen_threads, std::max(1u, h->gen_levels), std::max(1u, h->gen_frames)};
  h->spill.gen_overflow = h->gen_overflow.p;
  h->spill.gen_blocks_few = blocks_few;
  h->spill.gen_blocks_all = h->gen_threads / 256u >= blocks_all ? blocks_all : blocks_few;
}

// rpt_tree_generic raises a flag when a traversal outgrows its columns (they are sized from the scene, so that is a bug,
// not an input): read and cleared after every batch of queries — a render's and rptgpu_closest_hit's alike, so that the
// flag of one call never surfaces in the next.  The stream must be idle.
bool generic_overflowed(rptgpu_scene* h, hipStream_t st) {
  uint32_t flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, h->gen_overflow.p, sizeof flag, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (flag) {
    HIP_TRY(hipMemsetAsync(h->gen_overflow.p, 0, sizeof(uint32_t), st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return flag != 0;
}

void ensure_workspace(rptgpu_scene* h, uint64_t cap, uint32_t max_bounces) {
  if (cap <= h->ws_cap && max_bounces <= h->ws_bounces && h->ray.p) return;
  cap = std::max(cap, h->ws_cap);
  max_bounces = std::max(max_bounces, h->ws_bounces);
  int nl = std::max(1, h->dscene.num_lights);
  h->ray.alloc(6 * cap);
  h->hit.alloc(4 * cap);
  h->hit_obj.alloc(cap);
  h->draw.alloc(cap);
  h->nrec.alloc(cap);
  h->rec.release();
  h->rec.alloc((uint64_t)(max_bounces + 1) * rptdev::REC_FIELDS * cap);
  h->shadow.release();
  h->shadow.alloc((uint64_t)nl * rptdev::SHADOW_FIELDS * cap);
  h->queue_a.alloc(cap);
  h->queue_b.alloc(cap);
  h->counters.alloc(2 * (2 + (size_t)nl)); // two sets (rpt_shade clears the other one) of: [0] next-depth paths, [1] hits, [2 + l] shadow rays queued for light l
  h->shadow_q.release();
  h->shadow_q.alloc((uint64_t)nl * cap); // per light: the paths that cast a shadow ray towards it at the current depth
  h->srt.release();
  h->srt.alloc((uint64_t)nl * cap);      // per light and path: record.time of the shadow ray (rpt_shadow_sum reads it)
  if (h->has_deep) {
    h->tq.alloc(3 * cap); // a tree's ray queue | its rays with a zero direction component | those handed to the general form
    h->tq_ctr.alloc(16); // two sets of a tree's five counters, eight words apart (launch_query, QueryTuning::ctr_set)
    { // the traversal grid's stack spill area: one column per thread, as high as the scene's deepest tree (at least KD_MAX_STACK) less the LDS levels
      const uint64_t threads = (uint64_t)std::max(1, h->num_cus * 4) / 4 * RPT_TT_WAVES * 256;
      const uint64_t levels = (uint64_t)(std::max<uint32_t>((uint32_t)rptdev::KD_MAX_STACK, h->max_tree_depth + 1u) - RPT_TT_LEVELS_MIN);
      h->spill_node.alloc(levels * threads); h->spill_ts.alloc(levels * threads); h->spill_bmax.alloc(levels * threads);
      uint32_t zeros_common = 0; // every shadow ray towards an axis-parallel directional light has a zero component
      for (const rptdev::Light& l : h->host_lights)
        if (l.kind == RPT_LIGHT_DIRECTIONAL && (l.vec[0] == 0.0 || l.vec[1] == 0.0 || l.vec[2] == 0.0)) zeros_common = 1;
      h->tree_rays.alloc(8 * cap); // one 64-byte row per position of a query: the rays that enter a tree (StackSpill::rays)
      h->spill = StackSpill{h->spill_node.p, h->spill_ts.p, h->spill_bmax.p, (uint32_t)threads, zeros_common, h->tree_rays.p};
    }
    ensure_generic(h, h->gen_all);
    if (h->sort_rays) {
      h->sort_kin.alloc(cap); h->sort_kout.alloc(cap); h->sort_vin.alloc(cap);
      size_t bytes = rpt_strict::TABLE.sort_temp_bytes((uint32_t)cap);
      h->sort_tmp.alloc(bytes);
      h->sort_bufs = SortBufs{h->sort_kin.p, h->sort_kout.p, h->sort_vin.p, h->sort_tmp.p, bytes};
    }
  }
  h->ws_cap = cap;
  h->ws_bounces = max_bounces;
}

// -DRPT_PROF builds (kernels/prof.inc): per phase, the share of the waves' time, the lanes that were active while it
// ran (lane time / wave time, of 64), and for loop bodies the iteration count and the lanes per iteration.  One line
// per slot that was used, machine-readable enough to be committed under profiles/ as it is.
void print_prof(const KernelTable* kt, const char* what) {
  static const char* const NAMES[24] = {
      "tree_trace refill", "tree_trace node steps", "tree_trace box tests", "tree_trace pop", "tree_trace write-out",
      "tree_trace exact tests", "in-kernel node step", "in-kernel box batch", "in-kernel child test",
      "in-kernel triangle batch", "in-kernel object", "paths fetch", "paths raygen", "paths closest_hit",
      "paths illuminate", "paths visible", "paths nee_bsdf", "paths sample_f", "paths bsdf", "paths record",
      "paths fold+store", "flat candidate walk", "fold iteration", "rejection round"};
  unsigned long long t[4][24];
  if (!kt->read_prof(t)) return;
  unsigned long long tot = 0;
  for (int i = 0; i < 24; i++) tot += t[0][i];
  std::fprintf(stderr, "prof[%s] %-28s %8s %10s %14s %10s\n", what, "phase", "time %", "lanes/64", "iterations", "lanes/64");
  for (int i = 0; i < 24; i++) {
    if (!t[0][i] && !t[2][i]) continue;
    char a[32] = "-", b[32] = "-", c[32] = "-", d[32] = "-";
    if (t[0][i]) {
      std::snprintf(a, sizeof a, "%.2f", tot ? 100.0 * (double)t[0][i] / (double)tot : 0.0);
      std::snprintf(b, sizeof b, "%.1f", (double)t[1][i] / (double)t[0][i]);
    }
    if (t[2][i]) {
      std::snprintf(c, sizeof c, "%llu", t[2][i]);
      std::snprintf(d, sizeof d, "%.1f", (double)t[3][i] / (double)t[2][i]);
    }
    std::fprintf(stderr, "prof[%s] %-28s %8s %10s %14s %10s\n", what, NAMES[i], a, b, c, d);
  }
}

void release_workspace(rptgpu_scene* h) {
  h->ray.release(); h->hit.release(); h->hit_obj.release(); h->draw.release(); h->nrec.release(); h->rec.release();
  h->shadow.release(); h->queue_a.release(); h->queue_b.release(); h->tq.release(); h->srt.release(); h->shadow_q.release();
  h->sort_kin.release(); h->sort_kout.release(); h->sort_vin.release(); h->sort_tmp.release(); h->tree_rays.release();
  h->gen_defer.release(); h->gen_frame.release(); h->gen_threads = 0; // rpt_tree_generic's columns (ensure_generic makes them again)
  h->ws_cap = 0; h->ws_bounces = 0;
}

rptdev::Camera make_camera(const RptCamera& c) {
  // Camera::cast_ray derives d and right on every call (camera.rs:66-67); they are constants
  // of the batch, so they are computed once here with the same expressions.
  rptdev::Camera d{};
  std::memcpy(d.eye, c.eye, sizeof d.eye);
  std::memcpy(d.direction, c.direction, sizeof d.direction);
  std::memcpy(d.up, c.up, sizeof d.up);
  d.d = 1.0 / std::tan(c.fov / 2.0);
  const double* a = c.direction;
  const double* b = c.up;
  double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  double len = std::sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
  for (int k = 0; k < 3; k++) d.right[k] = cr[k] / len;
  d.aperture = c.aperture;
  d.focal_distance = c.focal_distance;
  return d;
}

// what is wrong with a batch's parameters (nullptr: nothing) — the same answer on every rank of a multi-GPU job
const char* bad_params(const RptRenderParams* p) {
  if (!p->width || !p->height || !p->iterations) return "width, height and iterations must be non-zero";
  if (p->max_bounces > 254) return "max_bounces > 254";
  if ((uint64_t)p->width * p->height >= (1ull << 31)) return "frame too large";
  if (p->part_count && p->part_index >= p->part_count) return "part_index >= part_count";
  if (p->precision_mode != RPT_PRECISION_F64_STRICT) return BAD_MODE;
  return nullptr;
}

// A handle whose aborted batch never drained (rptgpu_render_batch_reduce, drain_after_abort): the abandoned stream's
// kernels may still read and write the workspace, the frame buffers and events, so every call that would enqueue work
// refuses — until rptgpu_scene_destroy.
#define REFUSE_IF_ABANDONED(h)                                                                                            \
  do {                                                                                                                    \
    if ((h) && (h)->abandoned)                                                                                            \
      return fail((h), RPTGPU_E_COMM, "an aborted batch's device work never drained on this handle: destroy it (its "    \
                                      "workspace may still be written by the abandoned stream)");                        \
  } while (0)

// packed (with d_out, f32 or f64): d_out receives only this part's pixels, [npix][3] in the order of the part's pixel list
int render_impl(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* p, void* d_out, bool out_f32,
                double* host_out, hipStream_t user_stream, bool packed = false) {
  if (!h || !camera || !p) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null argument");
  REFUSE_IF_ABANDONED(h);
  if (const char* why = bad_params(p)) return fail(h, RPTGPU_E_INVALID_ARGUMENT, why);
  auto t0 = std::chrono::steady_clock::now();
  try {
    HIP_TRY(hipSetDevice(h->device));
    // hipGetLastError() reports the thread's LAST failed runtime call, whoever made it (another library in the process,
    // an unchecked clean-up call): start from a clean slate so that the checks below speak about this call's launches
    (void)hipGetLastError();
    hipStream_t st = h->stream;
    const KernelTable* kt = table_for(p->precision_mode, h->ext_shapes);
    const bool prof = (p->flags & RPT_FLAG_PROFILE_KERNELS) != 0;
    ensure_partition(h, *p);
    const uint32_t npix = h->npix;
    const uint64_t frame_elems = (uint64_t)p->width * p->height * 3;
    const size_t out_elem = out_f32 ? sizeof(float) : sizeof(double);
    void* out = d_out;
    if (!out) {
      h->out_full.alloc(frame_elems);
      out = h->out_full.p;
    }
    if (user_stream) HIP_TRY(hipStreamSynchronize(user_stream));
    if (!packed) HIP_TRY(hipMemsetAsync(out, 0, frame_elems * out_elem, st));
    // (a group with tree children is only walked by the per-tree kernels of the wavefront pipeline: RPT_FLAG_PERSISTENT
    // is a request such a scene cannot honour, not an error)
    const bool wavefront = (p->flags & RPT_FLAG_WAVEFRONT) || h->tree_kids ? true
                           : (p->flags & RPT_FLAG_PERSISTENT)             ? false
                                                                          : h->prefer_wavefront;
    h->dscene.force_general = (p->flags & RPT_FLAG_GENERAL_TRAVERSAL) ? 1 : 0;
    if (npix && !wavefront) {
      // ---- default pipeline: one persistent kernel, the whole path in registers
      h->accum.alloc((uint64_t)npix * 3);
      // a batch runs as n_launch launches of spp_l samples each, sized so one launch's per-sample radiance
      // buffer (24 B per sample) stays under lbuf_max_bytes: 512 spp at 1080p = 25.5 GB = one launch
      uint64_t lbuf_budget = h->lbuf_max_bytes;
      if (h->lbuf.n * sizeof(double) < lbuf_budget) { // growing: leave half of what is free to everyone else
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
          lbuf_budget = std::min<uint64_t>(lbuf_budget, std::max<uint64_t>(h->lbuf.n * sizeof(double), free_b / 2));
      }
      uint64_t spp_max = std::max<uint64_t>(1, lbuf_budget / ((uint64_t)npix * 3 * sizeof(double)));
      uint32_t n_launch = (uint32_t)(((uint64_t)p->iterations + spp_max - 1) / spp_max);
      uint32_t spp_l = n_launch ? (p->iterations + n_launch - 1) / n_launch : 0;
      // Samples per work item.  RptSceneOptions::paths_chunk = 0 (the default) chooses: 16 — 2 for flat scenes that run
      // the object filter AND trace long paths (max_bounces >= 4): there the lanes of a wave drift apart in path length
      // and short items keep a wave on one 8x8 pixel block and re-balance it often (the 23-polygon room at 8 bounces
      // 617 -> 664 Msamples/s, spheres.rs at 6 bounces 1899 -> 1981); with one or two segments per path every sample
      // costs the same and the per-item bookkeeping is all a short item adds (basic.rs 13418 -> 9391, the simple_video
      // frame 111 -> 88 frames/s at 2: profiles/r05_paths_chunk_ab.txt) — halved while a lane would get fewer
      // than 24 items: the launch's tail is one item long (a rank that owns an eighth of a 1080p frame at 128 spp:
      // x1.056 of the ideal 1/8 with 16 samples per item, x1.014 with 4; profiles/r05_emulated_ranks.txt).
      uint32_t chunk = h->paths_chunk;
      if (chunk == 0u) {
        chunk = (h->all_flat && !h->dscene.force_general && h->flat_layout.obj_filter && p->max_bounces >= 4u) ? 2u : 16u;
        const uint64_t lanes = (uint64_t)std::max(1, h->num_cus) * 8u * 64u;
        while (chunk > 1u && (uint64_t)npix * ((spp_l + chunk - 1) / chunk) < 24u * lanes) chunk /= 2u;
      }
      chunk = std::max(1u, std::min(chunk, std::max(1u, spp_l)));
      uint64_t n_items = (uint64_t)npix * ((spp_l + chunk - 1) / chunk);
      // 32-bit work counter: every lane of the grid may ask once past the end, and a wave's last guided claim may reach
      // past it (kernels/paths.inc fetch_item: at most 64 + 256 dead items per wave), so items + 8 x threads must fit
      // (slack: at most RPT_PATHS_WAVES_PER_CU_MAX one-wave blocks per CU — checked below — each with up to 64 askers past
      // the end and one last claim of at most RPT_PATHS_BATCH_MAX, the cap of a caller's paths_batch)
      const uint64_t item_limit = 0xFFFFFFF0ull - (uint64_t)h->num_cus * RPT_PATHS_WAVES_PER_CU_MAX * (64u + RPT_PATHS_BATCH_MAX);
      if (n_items > item_limit) {
        chunk = (uint32_t)(((uint64_t)spp_l * npix + item_limit - 1) / item_limit);
        while ((n_items = (uint64_t)npix * ((spp_l + chunk - 1) / chunk)) > item_limit) chunk++;
      }
      const bool flat = h->all_flat && !h->dscene.force_general;
      FlatLayout lay = flat ? h->flat_layout : FlatLayout{};
      const uint32_t flat_lds = lay.off_end;
      int per_cu = std::min(kt->paths_max_blocks_per_cu(flat ? &lay : nullptr, flat_lds, false), (int)RPT_PATHS_WAVES_PER_CU_MAX);
      // a texture environment: the lanes park their lookups in what the wave's LDS share has left (kernels/paths.inc) —
      // unless that costs a resident wave (a flat scene that fills the share)
      bool park = flat && h->opt.env_park != 0 && h->dscene.env_kind != RPT_ENV_COLOR; // (flat scenes: rpt_paths<KdLds>'s stack fills the share)
      if (park && kt->paths_max_blocks_per_cu(flat ? &lay : nullptr, flat_lds, true) < per_cu) park = false;
      uint32_t nblocks = (uint32_t)std::max(1, h->num_cus * per_cu);
      nblocks = (uint32_t)std::min<uint64_t>(nblocks, std::max<uint64_t>(1, (n_items + 63) / 64));
      uint64_t nthreads = (uint64_t)nblocks * 64;
      h->prec.alloc((uint64_t)rpt_fold_ring_slots(p->max_bounces) * rptdev::REC_FIELDS * nthreads);
      h->lbuf.alloc(std::max<uint64_t>(1, (uint64_t)spp_l * 3 * npix));
      if (std::getenv("RPTGPU_PRINT_LAUNCH"))
        std::fprintf(stderr, "rpt_paths<%s>: %d blocks/CU x %d CUs -> %u blocks, %u samples per work item, %u launch(es) of %u spp, "
                     "dynamic LDS %u B per wave (the flat scene's tables)%s\n",
                     flat ? (lay.obj_filter ? "KdFlatF" : lay.n_tris ? "KdFlat" : "KdFlatG") : "KdLds", per_cu, h->num_cus, nblocks, chunk, n_launch, spp_l, flat_lds,
                     park ? " + parked environment lookups" : "");
      h->counters.alloc(4);
      h->pcounters.alloc(16);
      HIP_TRY(hipMemsetAsync(h->pcounters.p, 0, 16 * sizeof(unsigned long long), st));
      rptdev::Frame fr{};
      fr.width = p->width; fr.height = p->height; fr.npix = npix; fr.pixels = h->pixels.p;
      fr.max_bounces = p->max_bounces; fr.seed = p->seed; fr.accum = h->accum.p;
      rptdev::Camera cam = make_camera(*camera);
      if (p->iterations == 0) HIP_TRY(hipMemsetAsync(h->accum.p, 0, (uint64_t)npix * 3 * sizeof(double), st));
      for (uint32_t s0 = 0; s0 < p->iterations; s0 += spp_l) {
        uint32_t spp = std::min(spp_l, p->iterations - s0);
        fr.sample_base = p->sample_index_base + s0;
        HIP_TRY(hipMemsetAsync(h->counters.p, 0, sizeof(uint32_t), st));
        { Bracket b(h, RPT_K_PATHS, prof);
          kt->paths(st, h->dscene, fr, cam, h->counters.p, h->prec.p, h->pcounters.p, h->lbuf.p, spp, chunk,
                    (uint32_t)((uint64_t)npix * ((spp + chunk - 1) / chunk)), nblocks, lay, flat, flat_lds, park, h->opt.paths_batch);
          b.done(); }
        kt->sum_samples(st, fr, h->lbuf.p, spp, s0 == 0);
      }
      HIP_TRY(hipGetLastError());
      kt->finish(st, fr, (double)p->iterations, std::pow(2.0, p->exposure_value), out, out_f32, packed);
      unsigned long long rc[16] = {0};
      HIP_TRY(hipMemcpyAsync(rc, h->pcounters.p, sizeof rc, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (std::getenv("RPTGPU_PRINT_PHASES")) print_prof(kt, "rpt_paths");
      h->stats.samples += (uint64_t)npix * p->iterations;
      h->stats.extend_rays += rc[0];
      h->stats.shadow_rays += rc[1];
      h->stats.shadow_rays_traced += rc[1]; // the persistent kernel traces every shadow ray (a skip there saves no wave time)
    } else if (npix) {
      // Paths in flight per pass.  Late bounces keep few paths alive, and a depth's kernels need ~10^5 rays to fill
      // 256 CUs, so the more paths start together the better the deep bounces run (C3 stand-in: 4 Mi -> 71, 16 Mi ->
      // 106, 128 Mi -> 128 Msamples/s; 16k-triangle glass 179 -> 324).  288 GB of HBM is what makes that affordable:
      // a path slot is ~0.8 KB at 8 bounces, so 128 Mi paths are ~100 GB of workspace.
      uint64_t target = h->target_paths;
      if (!target) {
        const uint64_t nl = (uint64_t)std::max(1, h->dscene.num_lights);
        uint64_t per_path = 6 * 8 + 4 * 8 + 4 + 4 + 1 + (uint64_t)(p->max_bounces + 1) * rptdev::REC_FIELDS * 8 +
                            nl * rptdev::SHADOW_FIELDS * 8 + 8 + nl * (8 + 4) + (h->has_deep ? 12 + 64 + (h->sort_rays ? 12 + 16 : 0) : 0);
        uint64_t budget = h->ws_budget_bytes;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
          uint64_t have = h->ws_cap * per_path; // what this handle already holds counts as available
          budget = std::min<uint64_t>(budget, (free_b + have) / 2);
        }
        target = std::min<uint64_t>(128ull << 20, std::max<uint64_t>(1ull << 20, budget / per_path));
      }
      uint32_t s_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(p->iterations, target / npix));
      // several handles (or processes) on one GPU each see the same `free` figure: if the pass does not fit after
      // all, halve it instead of failing the render (a smaller pass is only slower)
      // (rpt_tree_generic's large grid — whole objects, or under RPT_FLAG_GENERAL_TRAVERSAL everything, go through it: up
      // to several hundred MB of columns for a deep mesh — is part of the same attempt: if it does not fit, the pass shrinks)
      const bool generic_all = h->has_deep && (h->gen_all || h->dscene.force_general);
      for (;;) {
        try {
          ensure_workspace(h, (uint64_t)npix * s_chunk, p->max_bounces);
          if (generic_all) ensure_generic(h, true);
          break;
        } catch (const HipError& e) {
          if (e.e != hipErrorOutOfMemory || s_chunk == 1) throw;
          (void)hipGetLastError(); // clear the sticky error before retrying
          release_workspace(h);
          s_chunk = std::max(1u, s_chunk / 2);
        }
      }
      h->accum.alloc((uint64_t)npix * 3);
      HIP_TRY(hipMemsetAsync(h->accum.p, 0, (uint64_t)npix * 3 * sizeof(double), st));

      rptdev::PathState ps{};
      ps.ray = h->ray.p; ps.hit = h->hit.p; ps.hit_obj = h->hit_obj.p; ps.draw = h->draw.p;
      ps.nrec = h->nrec.p; ps.rec = h->rec.p; ps.shadow = h->shadow.p; ps.cap = h->ws_cap;
      rptdev::Frame fr{};
      fr.width = p->width; fr.height = p->height; fr.npix = npix; fr.pixels = h->pixels.p;
      fr.max_bounces = p->max_bounces; fr.seed = p->seed; fr.accum = h->accum.p;
      rptdev::Camera cam = make_camera(*camera);
      const bool any_lights = h->dscene.num_lights > 0;
      // the counter sets the kernels clear for each other start cleared (one memset per render, not one per depth and
      // per tree and query: 102 of the wine glass's 354 fills per step)
      const uint32_t nctr = 2u + (uint32_t)h->dscene.num_lights;
      HIP_TRY(hipMemsetAsync(h->counters.p, 0, 2 * (size_t)nctr * sizeof(uint32_t), st));
      uint32_t cset = 0;
      if (h->has_deep) {
        HIP_TRY(hipMemsetAsync(h->tq_ctr.p, 0, 16 * sizeof(uint32_t), st));
        h->qtune.ctr_set = 0;
      }
      QueryMarks qm(h, prof);
      const QueryHook qhook{query_mark, &qm};

      for (uint32_t s0 = 0; s0 < p->iterations; s0 += s_chunk) {
        uint32_t sc = std::min(s_chunk, p->iterations - s0);
        uint32_t n_paths = npix * sc;
        fr.sample_base = p->sample_index_base + s0;
        { Bracket b(h, RPT_K_RAYGEN, prof); kt->raygen(st, fr, cam, ps, n_paths); b.done(); }
        h->stats.samples += n_paths;
        uint32_t n_active = n_paths;
        const uint32_t* queue = nullptr; // identity at depth 0
        uint32_t* next = h->queue_a.p;
        for (uint32_t depth = 0; depth <= p->max_bounces && n_active; depth++) {
          // per-tree queries for scenes with deep trees; under RPT_FLAG_GENERAL_TRAVERSAL the whole scene is walked
          // in-kernel in the general form — unless it has a group with tree children, which only the per-tree pipeline
          // walks (there the flag sends every ray of every such object through rpt_tree_generic)
          const bool by_object = h->has_deep && (!(p->flags & RPT_FLAG_GENERAL_TRAVERSAL) || h->tree_kids);
          const uint32_t trace_blocks = (uint32_t)std::max(1, h->num_cus * 4);
          { Bracket b(h, RPT_K_EXTEND, prof);
            if (by_object)
              kt->query(st, h->dscene, ps, queue, n_active, -1, nullptr, nullptr, h->obj_deep.data(), h->obj_tris.data(),
                        h->dscene.num_objects, h->tq.p, h->tq_ctr.p, trace_blocks, h->sort_rays ? &h->sort_bufs : nullptr, &qhook, &h->spill, &h->qtune);
            else
              kt->extend(st, h->dscene, ps, queue, n_active);
            b.done(); }
          h->stats.extend_rays += n_active;
          const int nl = h->dscene.num_lights;
          uint32_t* const ctrs = h->counters.p + (size_t)cset * nctr;       // this depth's counters (cleared by the depth before)
          uint32_t* const ctrs_next = h->counters.p + (size_t)(cset ^ 1u) * nctr;
          cset ^= 1u;
          { Bracket b(h, RPT_K_SHADE, prof);
            kt->shade(st, h->dscene, fr, ps, queue, n_active, depth, next, ctrs, h->shadow_q.p, ctrs_next, nctr); b.done(); }
          // The depth's counts come back right after rpt_shade — the one point of a depth where the host waits — so the
          // visibility queries are sized for the shadow rays there ARE (50-70 % of the paths on closed meshes: less to
          // sort, smaller grids, and a light without a single ray at this depth costs no launch at all) and the next
          // depth for its survivors.  Until round 5 the wait stood at the depth's end and the queries ran over the
          // host's bound, the number of paths.  Everything up to the next rpt_shade is then enqueued without a wait.
          h->cnt_host.resize(2 + (size_t)nl);
          uint32_t* cnt = h->cnt_host.data();
          HIP_TRY(hipMemcpyAsync(cnt, ctrs, (2 + (size_t)nl) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
          for (int l = 0; l < nl; l++) h->stats.shadow_rays_traced += cnt[2 + l];
          if (prof && h->pending.size() >= 256) drain_events(h); // the stream is idle here: cheap
          h->stats.shadow_rays += (uint64_t)cnt[1] * (uint64_t)h->dscene.num_shadow_lights;
          if (any_lights) {
            // the visibility queries run over rpt_shade's per-light shadow-ray queues (their lengths also stay on the
            // device: ctrs + 2 + l is what the kernels read)
            Bracket b(h, RPT_K_SHADOW, prof);
            if (by_object) {
              for (int l = 0; l < nl; l++)
                if (h->light_casts[l] && cnt[2 + l])
                  kt->query(st, h->dscene, ps, h->shadow_q.p + (uint64_t)l * ps.cap, cnt[2 + l], l, h->srt.p, ctrs + 2 + l, h->obj_deep.data(), h->obj_tris.data(),
                            h->dscene.num_objects, h->tq.p, h->tq_ctr.p, trace_blocks, h->sort_rays ? &h->sort_bufs : nullptr, &qhook, &h->spill, &h->qtune);
            } else { // one launch for all lights of the depth (the grid's y is the light)
              uint32_t n_max = 0;
              for (int l = 0; l < nl; l++)
                if (h->light_casts[l]) n_max = std::max(n_max, cnt[2 + l]);
              if (n_max) kt->shadow_rays(st, h->dscene, ps, h->shadow_q.p, ctrs + 2, n_max, nl, h->srt.p);
            }
            kt->shadow_sum(st, h->dscene, ps, queue, n_active, depth, h->srt.p);
            b.done();
          }
          n_active = cnt[0];
          queue = next;
          next = (next == h->queue_a.p) ? h->queue_b.p : h->queue_a.p;
        }
        { Bracket b(h, RPT_K_RESOLVE, prof); kt->resolve(st, fr, ps, sc); b.done(); }
        HIP_TRY(hipGetLastError()); // a failed launch is reported here, not by the stream sync
      }
      kt->finish(st, fr, (double)p->iterations, std::pow(2.0, p->exposure_value), out, out_f32, packed);
      if (std::getenv("RPTGPU_PRINT_PHASES")) {
        HIP_TRY(hipStreamSynchronize(st));
        print_prof(kt, "wavefront");
      }
    }
    HIP_TRY(hipGetLastError());
    if (host_out) HIP_TRY(hipMemcpyAsync(host_out, out, frame_elems * out_elem, hipMemcpyDeviceToHost, st));
    uint32_t gen_overflow = 0; // (the flag rides with the call's last synchronisation; generic_overflowed() is the stand-alone form)
    if (wavefront && h->has_deep && h->gen_overflow.p)
      HIP_TRY(hipMemcpyAsync(&gen_overflow, h->gen_overflow.p, sizeof gen_overflow, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (prof) drain_events(h);
    if (gen_overflow) {
      (void)hipMemsetAsync(h->gen_overflow.p, 0, sizeof(uint32_t), st);
      return fail(h, RPTGPU_E_TREE_TOO_DEEP, "rpt_tree_generic: the traversal outgrew the stack sized for this scene (internal error)");
    }
  } catch (const HipError& e) {
    h->pending.clear();
    h->ev_used = 0;
    return hip_fail(h, e);
  } catch (const std::bad_alloc&) {
    return fail(h, RPTGPU_E_OUT_OF_MEMORY, "host allocation failed");
  } catch (...) {
    return fail(h, RPTGPU_E_HIP, "unexpected exception");
  }
  h->stats.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return RPTGPU_OK;
}

} // namespace

extern "C" {

int rptgpu_abi_version(void) { return RPTGPU_ABI_VERSION; }

const char* rptgpu_strerror(int code) {
  switch (code) {
    case RPTGPU_OK: return "ok";
    case RPTGPU_E_INVALID_ARGUMENT: return "invalid argument";
    case RPTGPU_E_UNSUPPORTED_SHAPE: return "shape outside the device's closed shape set";
    case RPTGPU_E_NO_DEVICE: return "no usable HIP device";
    case RPTGPU_E_HIP: return "HIP runtime error";
    case RPTGPU_E_OUT_OF_MEMORY: return "out of memory";
    case RPTGPU_E_TREE_TOO_DEEP: return "kd-tree deeper than the device traversal stack";
    case RPTGPU_E_UNIMPLEMENTED_SAMPLE: return "Shape::sample is unimplemented for this shape (plane.rs:34-36)";
    case RPTGPU_E_COMM: return "RCCL unavailable or collective failed";
    default: return "unknown error";
  }
}

const char* rptgpu_last_error_detail(const rptgpu_scene* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int rptgpu_device_count(int* out_count) {
  if (!out_count) return RPTGPU_E_INVALID_ARGUMENT;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *out_count = 0;
    return fail(nullptr, RPTGPU_E_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *out_count = n;
  return RPTGPU_OK;
}

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
  o->workspace_bytes = 96ull << 30;
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
    if (h->tree_kids) h->prefer_wavefront = true;
    h->all_flat = true;
    for (const rptdev::Tree& tr : fs.trees) h->all_flat = h->all_flat && tr.root_leaf != 0;
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

int rptgpu_render_batch(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params, double* out_rgb) {
  if (!out_rgb) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null out_rgb");
  return render_impl(h, camera, params, nullptr, false, out_rgb, nullptr);
}

int rptgpu_render_batch_device(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params, void* d_out,
                               int out_is_f32, void* stream) {
  if (!d_out) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null d_out");
  return render_impl(h, camera, params, d_out, out_is_f32 != 0, nullptr, (hipStream_t)stream);
}

int rptgpu_comm_unique_id(uint8_t out_id[RPTGPU_UNIQUE_ID_BYTES]) {
  if (!out_id) return RPTGPU_E_INVALID_ARGUMENT;
  Rccl& r = rccl();
  if (!r.ok) return fail(nullptr, RPTGPU_E_COMM, r.why);
  RcclUniqueId id;
  int rc = r.GetUniqueId(&id);
  if (rc != 0) return fail(nullptr, RPTGPU_E_COMM, std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
  static_assert(sizeof(id) == RPTGPU_UNIQUE_ID_BYTES, "unique id size");
  std::memcpy(out_id, &id, sizeof id);
  return RPTGPU_OK;
}

int rptgpu_comm_init(rptgpu_scene* h, int rank, int world, const uint8_t id[RPTGPU_UNIQUE_ID_BYTES]) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "bad rank / world / id");
  Rccl& r = rccl();
  if (!r.ok) return fail(h, RPTGPU_E_COMM, r.why);
  if (h->comm) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "the handle already has a communicator");
  REFUSE_IF_ABANDONED(h);
  h->comm_failed = false;
  if (hipSetDevice(h->device) != hipSuccess) return fail(h, RPTGPU_E_HIP, "hipSetDevice");
  RcclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  RcclComm c = nullptr;
  int rc = r.CommInitRank(&c, world, uid, rank);
  if (rc != 0) return fail(h, RPTGPU_E_COMM, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
  h->comm = c; h->comm_rank = rank; h->comm_world = world;
  return RPTGPU_OK;
}

int rptgpu_comm_destroy(rptgpu_scene* h) {
  if (!h) return RPTGPU_E_INVALID_ARGUMENT;
  if (h->comm && rccl().ok) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)rccl().CommDestroy(h->comm);
  }
  h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
  h->comm_failed = false;
  return RPTGPU_OK;
}

namespace {
// the root's view of the gather: every rank's pixel list (the lists the ranks' own ensure_partition builds) on the device
void ensure_gather_lists(rptgpu_scene* h, uint32_t width, uint32_t height, uint32_t world, uint32_t root) {
  uint32_t key[5] = {width, height, world, root, 1u};
  if (std::memcmp(key, h->gather_key, sizeof key) == 0 && h->gather_pixels.p) return;
  std::vector<uint32_t> all;
  all.reserve((size_t)width * height);
  h->gather_off.assign(world + 1, 0);
  for (uint32_t r = 0; r < world; r++) {
    std::vector<uint32_t> pix = pixel_list(width, height, 32, 8, r, world);
    all.insert(all.end(), pix.begin(), pix.end());
    h->gather_off[r + 1] = all.size();
  }
  h->gather_pixels.upload(all, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream)); // `all` dies with this function
  h->gather32.alloc(std::max<uint64_t>(1, all.size() * 3));
  std::memcpy(h->gather_key, key, sizeof key);
}
} // namespace

int rptgpu_render_batch_reduce(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params, int root,
                               float* out_rgb32) {
  if (!h || !camera || !params) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null argument");
  REFUSE_IF_ABANDONED(h);
  if (h->comm_failed)
    return fail(h, RPTGPU_E_COMM, "an earlier batch's collective failed on this handle: rptgpu_comm_destroy + rptgpu_comm_init before the next one");
  const int world = h->comm ? h->comm_world : 1, rank = h->comm ? h->comm_rank : 0;
  // errors every rank makes alike: returned before anything is enqueued, the communicator stays as it is
  if (root < 0 || root >= world) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "root out of range");
  if (const char* why = bad_params(params)) return fail(h, RPTGPU_E_INVALID_ARGUMENT, why);
  RptRenderParams p = *params;
  p.tile_width = 32; p.tile_height = 8; p.part_index = (uint32_t)rank; p.part_count = (uint32_t)world;
  const uint64_t n = (uint64_t)p.width * p.height * 3;
  Rccl* rc_lib = world > 1 ? &rccl() : nullptr;
  // From here on a failure is this rank's own (a null buffer on the root, out of memory, a HIP or RCCL error, a
  // time-out): the peers are in, or on their way into, the batch's collective and must not wait for this rank for
  // ever.  ncclCommAbort tears down this rank's side; the peers notice through ncclCommGetAsyncError or their own
  // time-out (wait_stream below) and do the same.  The handle then refuses further batches until it gets a new
  // communicator (comm_failed).
  auto abort_comm = [&] {
    if (world > 1 && h->comm) {
      if (rc_lib->CommAbort) (void)rc_lib->CommAbort(h->comm);
      h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
      h->comm_failed = true;
    }
  };
  auto fail_comm = [&](int code, const std::string& why) {
    abort_comm();
    return fail(h, code, why);
  };
  // After an abort the library's stream may still hold this batch's work — the scatter kernels and, on the root, the
  // copy into the CALLER's out_rgb32 — which the aborted collective now releases.  It is drained before the call
  // returns its error, so that nothing is written into the caller's buffer afterwards and the handle's next call finds
  // an idle stream.  Bounded: if the device does not finish within the communicator's time-out again (it should within
  // milliseconds once ncclCommAbort has returned), the handle gets a fresh stream, the old one is abandoned to the
  // runtime, and the error says that out_rgb32 may still be written to until the device is done.
  auto drain_after_abort = [&](std::string& note) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(h->opt.comm_timeout_s);
    for (uint32_t spins = 0;; spins++) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q != hipErrorNotReady) { if (q != hipSuccess) (void)hipGetLastError(); return; } // idle (or broken: nothing left to wait for)
      if (std::chrono::steady_clock::now() > deadline) break;
      if (spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    // The handle is finished: the old stream's kernels may still touch its workspace, frame buffers and events, so a
    // later render on a fresh stream would race with them.  Every call that enqueues work refuses from now on
    // (REFUSE_IF_ABANDONED); rptgpu_scene_destroy frees the memory (hipFree waits for the device).
    h->abandoned = true;
    note = " — the library's stream did not drain after the abort: out_rgb32 may be written to until the device finishes, and this handle accepts no further work (destroy it)";
  };
  if (rank == root && !out_rgb32) return fail_comm(RPTGPU_E_INVALID_ARGUMENT, "null out_rgb32 on the root rank");
  // waits for the library's stream; with a communicator it polls the stream together with RCCL's asynchronous error
  // state instead of blocking, so that a peer's failure ends this call too
  auto wait_stream = [&]() -> int {
    if (world <= 1) { HIP_TRY(hipStreamSynchronize(h->stream)); return RPTGPU_OK; }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(h->opt.comm_timeout_s);
    for (uint32_t spins = 0;; spins++) {
      hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) return RPTGPU_OK;
      if (q != hipErrorNotReady) throw HipError{q, "hipStreamQuery", __LINE__};
      if (rc_lib->CommGetAsyncError) {
        int aerr = 0;
        int grc = rc_lib->CommGetAsyncError(h->comm, &aerr);
        if (grc != 0 || (aerr != 0 && aerr != RCCL_IN_PROGRESS)) {
          const int code = grc != 0 ? grc : aerr;
          abort_comm();
          std::string note;
          drain_after_abort(note);
          return fail(h, RPTGPU_E_COMM, std::string("asynchronous RCCL error while the batch's collective was in flight: ") +
                                            (rc_lib->GetErrorString ? rc_lib->GetErrorString(code) : "error") + note);
        }
      }
      if (std::chrono::steady_clock::now() > deadline) {
        abort_comm();
        std::string note;
        drain_after_abort(note);
        return fail(h, RPTGPU_E_COMM, "the batch's collective did not finish within RptSceneOptions::comm_timeout_s (a peer rank failed or hangs)" + note);
      }
      if (spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  };
  if (params->collective > RPT_COLLECTIVE_REDUCE) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "RptRenderParams::collective");
  bool gather = params->collective != RPT_COLLECTIVE_REDUCE;
  if (const char* mode_env = std::getenv("RPTGPU_COLLECTIVE")) { // (an override for experiments: every rank sees the same environment)
    if (std::strcmp(mode_env, "reduce") == 0) gather = false;
    else if (std::strcmp(mode_env, "gather") == 0) gather = true;
  }
  if (world > 1 && gather && !(rc_lib->Send && rc_lib->Recv && rc_lib->GroupStart && rc_lib->GroupEnd)) gather = false;
  if (!h->comm) gather = false; // no communicator: a plain render straight into the frame (a 1-rank communicator still packs and places)
  try {
    HIP_TRY(hipSetDevice(h->device));
    for (auto& e : h->ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
    if (rank == root && (gather || world > 1)) h->frame32_sum.alloc(n);
    if (gather) {
      ensure_partition(h, p);
      h->packed32.alloc(std::max<uint64_t>(1, (uint64_t)h->npix * 3));
      if (rank == root) ensure_gather_lists(h, p.width, p.height, (uint32_t)world, (uint32_t)root);
    } else {
      h->frame32.alloc(n);
    }
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  } catch (const HipError& e) {
    abort_comm();
    return hip_fail(h, e);
  } catch (const std::bad_alloc&) {
    return fail_comm(RPTGPU_E_HIP, "out of host memory");
  }
  // this rank's tiles: gather — only its pixels, packed; reduce — the full frame with zeros elsewhere
  int rc = gather ? render_impl(h, camera, &p, h->packed32.p, true, nullptr, nullptr, true)
                  : render_impl(h, camera, &p, h->frame32.p, true, nullptr, nullptr);
  if (rc != RPTGPU_OK) {
    std::string detail = h->error; // keep the render's own message
    abort_comm();
    h->error = detail;
    return rc;
  }
  try {
    const KernelTable* kt = table_for(p.precision_mode, h->ext_shapes);
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    float* result = nullptr;
    if (gather) {
      if (world > 1) {
        int nrc = rc_lib->GroupStart();
        if (nrc == 0) {
          if (rank == root) {
            for (int r = 0; r < world && nrc == 0; r++) {
              if (r == root) continue;
              const uint64_t cnt = (h->gather_off[r + 1] - h->gather_off[r]) * 3;
              if (cnt) nrc = rc_lib->Recv(h->gather32.p + h->gather_off[r] * 3, (size_t)cnt, RCCL_FLOAT32, r, h->comm, h->stream);
            }
          } else if (h->npix) {
            nrc = rc_lib->Send(h->packed32.p, (size_t)h->npix * 3, RCCL_FLOAT32, root, h->comm, h->stream);
          }
          const int erc = rc_lib->GroupEnd();
          if (nrc == 0) nrc = erc;
        }
        if (nrc != 0) {
          (void)hipStreamSynchronize(h->stream);
          return fail_comm(RPTGPU_E_COMM, std::string("ncclSend / ncclRecv: ") + (rc_lib->GetErrorString ? rc_lib->GetErrorString(nrc) : "error"));
        }
      }
      HIP_TRY(hipEventRecord(h->ev[2], h->stream));
      if (rank == root) { // every rank's pixels into their places; between them the lists cover the frame exactly once
        for (int r = 0; r < world; r++) {
          const uint64_t off = h->gather_off[r], cnt = h->gather_off[r + 1] - off;
          kt->scatter_f32(h->stream, r == root ? h->packed32.p : h->gather32.p + off * 3, h->gather_pixels.p + off, (uint32_t)cnt, h->frame32_sum.p);
        }
        HIP_TRY(hipGetLastError());
        result = h->frame32_sum.p;
      }
    } else {
      result = h->frame32.p;
      if (world > 1) {
        int nrc = rc_lib->Reduce(h->frame32.p, rank == root ? h->frame32_sum.p : nullptr, (size_t)n, RCCL_FLOAT32, RCCL_SUM, root, h->comm, h->stream);
        if (nrc != 0) {
          (void)hipStreamSynchronize(h->stream);
          return fail_comm(RPTGPU_E_COMM, std::string("ncclReduce: ") + (rc_lib->GetErrorString ? rc_lib->GetErrorString(nrc) : "error"));
        }
        result = h->frame32_sum.p;
      }
      HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    }
    if (rank == root) HIP_TRY(hipMemcpyAsync(out_rgb32, result, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    if (int wrc = wait_stream(); wrc != RPTGPU_OK) return wrc;
    float ms[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; k++) HIP_TRY(hipEventElapsedTime(&ms[k], h->ev[k], h->ev[k + 1]));
    h->stats.reduce_calls += 1;
    h->stats.reduce_render_ms += ms[0];
    h->stats.reduce_collective_ms += ms[1];
    h->stats.reduce_copy_ms += ms[2];
  } catch (const HipError& e) {
    abort_comm();
    std::string note;
    if (world > 1) drain_after_abort(note);
    return hip_fail(h, e);
  } catch (const std::bad_alloc&) {
    return fail_comm(RPTGPU_E_HIP, "out of host memory");
  }
  return RPTGPU_OK;
}

int rptgpu_render_batch_emulate_ranks(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params, int world,
                                      float* out_rgb32) {
  if (!h || !camera || !params || !out_rgb32) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null argument");
  REFUSE_IF_ABANDONED(h);
  if (world < 1 || world > 4096) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "world out of range");
  if (const char* why = bad_params(params)) return fail(h, RPTGPU_E_INVALID_ARGUMENT, why);
  const uint64_t n = (uint64_t)params->width * params->height * 3;
  try {
    HIP_TRY(hipSetDevice(h->device));
    h->frame32_sum.alloc(n);
    ensure_gather_lists(h, params->width, params->height, (uint32_t)world, 0u);
    HIP_TRY(hipMemsetAsync(h->frame32_sum.p, 0xff, n * sizeof(float), h->stream)); // NaNs: a pixel nobody places shows
  } catch (const HipError& e) {
    return hip_fail(h, e);
  } catch (const std::bad_alloc&) { // (ensure_gather_lists builds host lists of width * height entries)
    return fail(h, RPTGPU_E_OUT_OF_MEMORY, "host allocation failed");
  }
  // what each rank would send, rendered here one after the other straight into the root's receive buffer
  for (int r = 0; r < world; r++) {
    RptRenderParams p = *params;
    p.tile_width = 32; p.tile_height = 8; p.part_index = (uint32_t)r; p.part_count = (uint32_t)world;
    if (h->gather_off[r + 1] == h->gather_off[r]) continue; // a rank without a tile (more ranks than tiles)
    int rc = render_impl(h, camera, &p, h->gather32.p + h->gather_off[r] * 3, true, nullptr, nullptr, true);
    if (rc != RPTGPU_OK) return rc;
  }
  try {
    const KernelTable* kt = table_for(params->precision_mode, h->ext_shapes);
    for (int r = 0; r < world; r++) {
      const uint64_t off = h->gather_off[r], cnt = h->gather_off[r + 1] - off;
      kt->scatter_f32(h->stream, h->gather32.p + off * 3, h->gather_pixels.p + off, (uint32_t)cnt, h->frame32_sum.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rgb32, h->frame32_sum.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (const HipError& e) {
    return hip_fail(h, e);
  }
  return RPTGPU_OK;
}

int rptgpu_closest_hit(rptgpu_scene* h, uint64_t n, const double* origins, const double* dirs,
                       uint32_t precision_mode, double* out_t, double* out_normal, int32_t* out_object) {
  if (!h || (n && (!origins || !dirs || !out_t || !out_normal || !out_object)))
    return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null argument");
  if (precision_mode != RPT_PRECISION_F64_STRICT) return fail(h, RPTGPU_E_INVALID_ARGUMENT, BAD_MODE);
  REFUSE_IF_ABANDONED(h);
  if (!n) return RPTGPU_OK;
  try {
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    if (h->has_deep && (!h->rays_in_kernel || h->tree_kids)) {
      // a scene with deep trees: the rays take the route a render's rays take — object by object, every deep tree with
      // its own queue, sort and persistent traversal (launch_query) — in pieces of at most 4 Mi rays
      const KernelTable* kt = table_for(precision_mode, h->ext_shapes);
      const uint64_t piece = std::min<uint64_t>(n, 4ull << 20);
      ensure_workspace(h, piece, 0);
      rptdev::PathState ps{};
      ps.ray = h->ray.p; ps.hit = h->hit.p; ps.hit_obj = h->hit_obj.p; ps.draw = h->draw.p;
      ps.nrec = h->nrec.p; ps.rec = h->rec.p; ps.shadow = h->shadow.p; ps.cap = h->ws_cap;
      const uint32_t trace_blocks = (uint32_t)std::max(1, h->num_cus * 4);
      std::vector<double> soa(6 * piece), hit(4 * piece);
      for (uint64_t base = 0; base < n; base += piece) {
        const uint64_t m = std::min(piece, n - base);
        for (uint64_t i = 0; i < m; i++)
          for (int k = 0; k < 3; k++) {
            soa[(uint64_t)k * m + i] = origins[3 * (base + i) + k];
            soa[(uint64_t)(3 + k) * m + i] = dirs[3 * (base + i) + k];
          }
        for (int k = 0; k < 6; k++)
          HIP_TRY(hipMemcpyAsync(ps.ray + (uint64_t)k * ps.cap, soa.data() + (uint64_t)k * m, m * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(h->tq_ctr.p, 0, 16 * sizeof(uint32_t), st));
        h->qtune.ctr_set = 0;
        kt->query(st, h->dscene, ps, nullptr, (uint32_t)m, -1, nullptr, nullptr, h->obj_deep.data(), h->obj_tris.data(),
                  h->dscene.num_objects, h->tq.p, h->tq_ctr.p, trace_blocks, h->sort_rays ? &h->sort_bufs : nullptr, nullptr, &h->spill, &h->qtune);
        HIP_TRY(hipGetLastError());
        for (int k = 0; k < 4; k++)
          HIP_TRY(hipMemcpyAsync(hit.data() + (uint64_t)k * m, ps.hit + (uint64_t)k * ps.cap, m * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_object + base, ps.hit_obj, m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < m; i++) {
          out_t[base + i] = hit[i];
          for (int k = 0; k < 3; k++) out_normal[3 * (base + i) + k] = hit[(uint64_t)(1 + k) * m + i];
        }
      }
      if (h->gen_overflow.p && generic_overflowed(h, st))
        return fail(h, RPTGPU_E_TREE_TOO_DEEP, "rpt_tree_generic: the traversal outgrew the stack sized for this scene (internal error)");
      return RPTGPU_OK;
    }
    DevBuf<double> d_o, d_d, d_t, d_n;
    DevBuf<int32_t> d_obj;
    d_o.alloc(3 * n); d_d.alloc(3 * n); d_t.alloc(n); d_n.alloc(3 * n); d_obj.alloc(n);
    HIP_TRY(hipMemcpyAsync(d_o.p, origins, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_d.p, dirs, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
    table_for(precision_mode, h->ext_shapes)->extend_rays(st, h->dscene, d_o.p, d_d.p, n, d_t.p, d_n.p, d_obj.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_t, d_t.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_normal, d_n.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_object, d_obj.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  } catch (const HipError& e) {
    return hip_fail(h, e);
  } catch (...) {
    return fail(h, RPTGPU_E_HIP, "unexpected exception");
  }
  return RPTGPU_OK;
}

int rptgpu_eval_math(rptgpu_scene* h, int fn, uint64_t n, const double* x, const double* y, double* out) {
  if (!h || (n && (!x || !out)) || fn < 0 || fn > 7 || (fn >= 6 && n && !y))
    return fail(h, RPTGPU_E_INVALID_ARGUMENT, "bad argument");
  if (!n) return RPTGPU_OK;
  DevBuf<double> dx, dy, dout;
  int rc = RPTGPU_OK;
  try {
    HIP_TRY(hipSetDevice(h->device));
    dx.alloc(n); dy.alloc(n); dout.alloc(n);
    HIP_TRY(hipMemcpyAsync(dx.p, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (y) HIP_TRY(hipMemcpyAsync(dy.p, y, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    else HIP_TRY(hipMemsetAsync(dy.p, 0, n * sizeof(double), h->stream));
    rpt_strict::TABLE.eval_math(h->stream, fn, n, dx.p, dy.p, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (const HipError& e) {
    rc = hip_fail(h, e);
  } catch (...) {
    rc = fail(h, RPTGPU_E_HIP, "unexpected exception");
  }
  dx.release(); dy.release(); dout.release();
  return rc;
}

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

// ------------------------------------------------------------------ device-resident Buffer
struct rptgpu_buffer {
  rptgpu_scene* h = nullptr;
  uint32_t width = 0, height = 0, radius = 0;
  std::vector<double*> batches; // one W*H*3 frame per add_samples call, on the device
  DevBuf<double> total, thr, pix_var;
  DevBuf<const double*> batch_ptrs;
  DevBuf<uint8_t> image;
};

namespace {
// color_bytes (color.rs:18-24) as the host computes it; the device reproduces it from thresholds
inline int color_byte_host(double v) {
  double t = std::pow(std::fmin(std::fmax(v, 0.0), 1.0), 1.0 / 2.2) * 255.0;
  return !(t > 0.0) ? 0 : (t >= 255.0 ? 255 : (int)t);
}
// smallest v in [0,1] with color_byte_host(v) >= k, by bisection over the doubles; `clean` reports
// whether the conversion is a step function in a window of +-256 ulps around every threshold
std::vector<double> byte_thresholds(bool& clean) {
  std::vector<double> thr(256, 0.0);
  clean = true;
  for (int k = 1; k < 256; k++) {
    uint64_t lo = 0, hi;
    double one = 1.0;
    std::memcpy(&hi, &one, 8); // positive doubles order like their bit patterns
    while (lo < hi) {
      uint64_t mid = lo + (hi - lo) / 2;
      double v;
      std::memcpy(&v, &mid, 8);
      if (color_byte_host(v) >= k) hi = mid;
      else lo = mid + 1;
    }
    std::memcpy(&thr[k], &lo, 8);
    for (int d = -256; d <= 256; d++) {
      uint64_t u = lo + (uint64_t)(int64_t)d;
      double v;
      std::memcpy(&v, &u, 8);
      if (v >= 0.0 && v <= 1.0 && (color_byte_host(v) >= k) != (d >= 0)) clean = false;
    }
  }
  return thr;
}
} // namespace

extern "C" {

int rptgpu_buffer_create(rptgpu_scene* h, uint32_t width, uint32_t height, uint32_t filter_radius, rptgpu_buffer** out) {
  if (!h || !out || !width || !height) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "bad argument");
  *out = nullptr;
  rptgpu_buffer* b = new (std::nothrow) rptgpu_buffer();
  if (!b) return fail(h, RPTGPU_E_OUT_OF_MEMORY, "host allocation failed");
  b->h = h; b->width = width; b->height = height; b->radius = filter_radius;
  try {
    HIP_TRY(hipSetDevice(h->device));
    bool clean = true;
    std::vector<double> thr = byte_thresholds(clean);
    if (!clean) {
      delete b;
      return fail(h, RPTGPU_E_INVALID_ARGUMENT, "host pow() is not monotone around a u8 threshold");
    }
    b->thr.upload(thr, h->stream);
    uint64_t n = (uint64_t)width * height * 3;
    b->total.alloc(n);
    HIP_TRY(hipMemsetAsync(b->total.p, 0, n * sizeof(double), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (const HipError& e) {
    int code = hip_fail(h, e);
    delete b;
    return code;
  }
  *out = b;
  return RPTGPU_OK;
}

void rptgpu_buffer_destroy(rptgpu_buffer* b) {
  if (!b) return;
  (void)hipSetDevice(b->h->device);
  for (double* p : b->batches) (void)hipFree(p);
  b->total.release(); b->thr.release(); b->pix_var.release(); b->batch_ptrs.release(); b->image.release();
  delete b;
}

int rptgpu_buffer_sample(rptgpu_buffer* b, const RptCamera* camera, const RptRenderParams* params) {
  if (!b || !camera || !params) return RPTGPU_E_INVALID_ARGUMENT;
  rptgpu_scene* h = b->h;
  if (params->width != b->width || params->height != b->height)
    return fail(h, RPTGPU_E_INVALID_ARGUMENT, "Invalid sample dimension"); // buffer.rs:33-36
  double* frame = nullptr;
  uint64_t n = (uint64_t)b->width * b->height * 3;
  if (hipSetDevice(h->device) != hipSuccess || hipMalloc((void**)&frame, n * sizeof(double)) != hipSuccess)
    return fail(h, RPTGPU_E_OUT_OF_MEMORY, "hipMalloc of a batch frame failed");
  int rc = render_impl(h, camera, params, frame, false, nullptr, nullptr);
  if (rc != RPTGPU_OK) {
    (void)hipFree(frame);
    return rc;
  }
  try {
    table_for(RPT_PRECISION_F64_STRICT)->buffer_add(h->stream, b->total.p, frame, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (const HipError& e) {
    (void)hipFree(frame);
    return hip_fail(h, e);
  }
  b->batches.push_back(frame);
  return RPTGPU_OK;
}

int rptgpu_buffer_image(rptgpu_buffer* b, uint8_t* out_rgb8) {
  if (!b || !out_rgb8) return RPTGPU_E_INVALID_ARGUMENT;
  rptgpu_scene* h = b->h;
  if (b->batches.empty()) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "Pixel found with no samples"); // buffer.rs:89
  try {
    HIP_TRY(hipSetDevice(h->device));
    uint64_t n = (uint64_t)b->width * b->height * 3;
    b->image.alloc(n);
    table_for(RPT_PRECISION_F64_STRICT)->buffer_image(h->stream, b->total.p, b->width, b->height, b->radius,
                                                      (uint32_t)b->batches.size(), b->thr.p, b->image.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rgb8, b->image.p, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (const HipError& e) {
    return hip_fail(h, e);
  }
  return RPTGPU_OK;
}

int rptgpu_buffer_variance(rptgpu_buffer* b, double* out_variance) {
  if (!b || !out_variance) return RPTGPU_E_INVALID_ARGUMENT;
  rptgpu_scene* h = b->h;
  if (b->batches.empty()) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "no samples");
  try {
    HIP_TRY(hipSetDevice(h->device));
    uint64_t npix = (uint64_t)b->width * b->height;
    std::vector<const double*> ptrs(b->batches.begin(), b->batches.end());
    b->batch_ptrs.upload(ptrs, h->stream);
    b->pix_var.alloc(npix);
    table_for(RPT_PRECISION_F64_STRICT)->buffer_variance(h->stream, b->total.p, b->batch_ptrs.p,
                                                         (uint32_t)b->batches.size(), npix, b->pix_var.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> pv(npix);
    HIP_TRY(hipMemcpyAsync(pv.data(), b->pix_var.p, npix * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double variance = 0.0, count = 0.0; // buffer.rs:60-72: sequential sum over pixels in index order
    for (uint64_t p = 0; p < npix; p++) {
      variance += pv[p];
      count += 1.0;
    }
    *out_variance = variance / count;
  } catch (const HipError& e) {
    return hip_fail(h, e);
  } catch (...) {
    return fail(h, RPTGPU_E_OUT_OF_MEMORY, "host allocation failed");
  }
  return RPTGPU_OK;
}

int rptgpu_buffer_num_batches(const rptgpu_buffer* b, uint32_t* out) {
  if (!b || !out) return RPTGPU_E_INVALID_ARGUMENT;
  *out = (uint32_t)b->batches.size();
  return RPTGPU_OK;
}

} // extern "C" (buffer)

extern "C" {

int rptgpu_get_stats(const rptgpu_scene* h, RptStats* out) {
  if (!h || !out) return RPTGPU_E_INVALID_ARGUMENT;
  *out = h->stats;
  return RPTGPU_OK;
}

int rptgpu_reset_stats(rptgpu_scene* h) {
  if (!h) return RPTGPU_E_INVALID_ARGUMENT;
  std::memset(&h->stats, 0, sizeof h->stats);
  return RPTGPU_OK;
}

const char* rptgpu_kernel_name(int k) {
  switch (k) {
    case RPT_K_RAYGEN: return "rpt_raygen";
    case RPT_K_EXTEND: return "rpt_extend";
    case RPT_K_SHADE: return "rpt_shade";
    case RPT_K_SHADOW: return "rpt_shadow"; // the visibility queries of a depth: rpt_shadow_rays or the per-tree kernels, + rpt_shadow_sum
    case RPT_K_RESOLVE: return "rpt_resolve";
    case RPT_K_PATHS: return "rpt_paths";
    case RPT_K_TREE_TRACE: return "rpt_tree_trace";
    case RPT_K_TREE_SORT: return "rpt_tree_enter+sort";
    default: return "";
  }
}

} // extern "C"
