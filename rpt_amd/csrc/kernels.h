// kernels.h — host-callable launchers of the gfx950 kernels, one table per arithmetic mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

// meshes per batched run of the flat path kernel (kernels/paths.inc flat_query; api_scene.cpp caps Inst::plane_use with it)
#ifndef RPT_FLAT_RUN
#define RPT_FLAT_RUN 6
#endif

// slots (of REC_FIELDS doubles) per thread of the persistent path kernel's record ring (kernels/paths.inc says why)
static inline uint32_t rpt_fold_ring_slots(uint32_t max_bounces) { return 3u * max_bounces + 2u; }
// LDS of one wave of rpt_paths: 160 KB per CU / (2 waves per SIMD x 4 SIMDs); the fold walker's static share of it
#define RPT_PATHS_WAVE_LDS 20480u
#define RPT_PATHS_WALKER_LDS 2560u
// rpt_paths<KdFlat> (a flat scene WITH its triangles in LDS) also keeps the lanes' stashed camera rays there
#define RPT_PATHS_STASH_LDS 4864u
// flat scenes with a texture environment: the lanes' queues of parked lookups, RPT_PARK_K entries each, at the end of the
// wave's dynamic LDS (kernels/paths.inc ParkLds)
#ifndef RPT_PARK_K
#define RPT_PARK_K 4
#endif
#define RPT_PATHS_PARK_LDS (RPT_PARK_K * 2624u)
// rpt_paths's 32-bit work counter: a wave's last claim may overshoot the end by its batch (kernels/paths.inc fetch_item), so
// a caller's RptSceneOptions::paths_batch is capped, and api_render.cpp leaves WAVES_PER_CU_MAX x (64 + BATCH_MAX) items of room
// per CU (one-wave blocks: the occupancy query's answer is clamped to the same bound)
// the wavefront pipeline's passes: at most this many paths in flight (32-bit slot indices, queue counters and grids have
// room for 2^31), and at most this share of the device's free memory for their state
#ifndef RPT_MAX_PATHS_PER_PASS
#define RPT_MAX_PATHS_PER_PASS (512ull << 20)
#endif
#ifndef RPT_WS_FREE_PERCENT
#define RPT_WS_FREE_PERCENT 85
#endif
// the paths of a depth are re-ordered by ray key in scenes without per-tree queues when there are at least this many
#ifndef RPT_PATH_REORDER_MIN
#define RPT_PATH_REORDER_MIN (1u << 20)
#endif
#define RPT_PATHS_BATCH_MAX 1024u
#define RPT_PATHS_WAVES_PER_CU_MAX 32u

// layout of the flat path kernel's dynamic LDS (byte offsets; lrec at 0), see kernels.inc
struct FlatLayout {
  uint32_t off_tris, off_refs, off_mat, off_leaf;
  uint32_t off_end;              // end of the scene's tables (0 for scenes that are not flat)
  uint32_t n_refs, n_tris;
  // distinct bounding-plane coordinates of the untransformed meshes, at most 4 per axis: the quotient
  // (value - o) / d of each is computed ONCE per ray into off_qtab ([distinct planes][64 lanes] doubles) and shared
  // by every mesh whose box uses that plane (the walls of C2 have 30 faces on 6 distinct planes)
  uint32_t off_qtab, plane_cnt;  // plane_cnt: 4 bits per axis; 0 = feature off
  const double* plane_vals;      // [3][4] in device memory
  // the object filter of flat scenes with many objects and no plane table (kernels/paths.inc flat_query_filtered): a
  // conservative 16-bit box per top-level object on a grid over all of them, tested in f32 before the object's own
  // (exact) test; bit k of obj_always = object k is never filtered (a Plane, a mesh with a sliver, ...)
  uint32_t obj_filter;           // 0 = off
  uint32_t off_obox;             // [objects][6] doubles in LDS: the bounds of MESH objects (their exact slab test)
  const rptdev::LeafBox* obj_box; // [objects] in device memory
  const double* obj_grid;        // qlo[3], qscale[3], bounds[6] of the grid, device memory
  uint64_t obj_always;
};

// buffers of the optional ray sort in front of a per-tree traversal (all sized for the query's n)
struct SortBufs {
  uint32_t *keys_in, *keys_out, *vals_in;
  void* tmp;
  size_t tmp_bytes;
};

// waves per SIMD of the per-tree traversal kernels and the stack levels they keep in LDS
// (RPT_TT_WAVES * 4 * 64 lanes * 20 B * levels <= 160 KB per CU); deeper levels go to the spill area
#ifndef RPT_TT_WAVES
#define RPT_TT_WAVES 4
#endif
#ifndef RPT_TT_LEVELS
#define RPT_TT_LEVELS 7
#endif
#define RPT_TT_LEVELS_MIN RPT_TT_LEVELS
// spill area of the traversal stack beyond the LDS levels: [KD_MAX_STACK - RPT_TT_LEVELS_MIN][threads] per array, one
// column per thread of the traversal grid (api_render.cpp allocates it for scenes with deep trees)
// rpt_tree_generic's pending work (kernels/wavefront.inc), one column per thread of ITS grid: deferred far children
// (six face parameters, t_split, node) and the suspended leaves of the groups above the tree being walked
struct GenericStack {
  double* defer;  // [levels][8][threads]
  double* frame;  // [frames][12][threads]
  uint32_t threads, levels, frames;
};
struct StackSpill {
  uint32_t* node;
  double* ts;
  double* bmax;
  uint32_t threads;
  uint32_t zeros_common; // scene hint for launch_query: rays with a zero direction component are frequent (full grid for their kernel)
  // [position in the query][8]: object-space origin and direction, record.time and stop distance of a ray that enters
  // the tree — written by rpt_tree_enter, which has them in registers, as ONE 64-byte row; the traversal kernels' refill
  // reads that row (and the slot from the query's queue) instead of gathering eight values from the path state's SoA
  // arrays.  The tree's queues hold positions, not slots.
  double* rays;
  // rpt_tree_generic: its columns, the flag it raises if a scene outgrows them (api_render.cpp sizes them from the scene and
  // checks the flag when the batch is done), and its grid when it takes a few handed-on rays / every ray of an object
  GenericStack gen;
  uint32_t* gen_overflow;
  uint32_t gen_blocks_few, gen_blocks_all;
};

// accounting hook of launch_query: called with (ctx, kind, 0) before and (ctx, kind, 1) after the launches of
// a phase of the query (kind = RPT_K_TREE_TRACE / RPT_K_TREE_SORT); may be null
struct QueryHook {
  void (*mark)(void* ctx, int kind, int end);
  void* ctx;
};

// launch_query's state and tuning, owned by the caller (one per handle): which of the two sets of tree counters the
// next (tree, query) pair uses — both sets cleared when the buffer is made and after a failed render — and the
// smallest query that is still sorted (RptSceneOptions::sort_min_rays)
struct QueryTuning {
  uint32_t ctr_set;
  uint32_t sort_min_rays;
};

struct KernelTable {
  void (*raygen)(hipStream_t, const rptdev::Frame&, const rptdev::Camera&, const rptdev::PathState&, uint32_t n_paths);
  void (*extend)(hipStream_t, const rptdev::Scene&, const rptdev::PathState&, const uint32_t* queue, uint32_t n);
  void (*extend_rays)(hipStream_t, const rptdev::Scene&, const double* o, const double* d, uint64_t n, double* out_t,
                      double* out_n, int32_t* out_obj);
  // counters: [0] paths of the next depth, [1] hits, [2 + l] shadow rays queued for light l; sq: those queues ([light][cap])
  void (*shade)(hipStream_t, const rptdev::Scene&, const rptdev::Frame&, const rptdev::PathState&,
                const uint32_t* queue, uint32_t n, uint32_t depth, uint32_t* next_queue, uint32_t* counters, uint32_t* sq,
                uint32_t* zero_next /* rpt_shade clears these zero_n words: the next depth's counters */, uint32_t zero_n,
                uint32_t rec_off /* the depth's first record column (PathState::rec) */);
  // visibility of every light of the depth over its shadow-ray queue ([light][cap], lengths sq_counts[light] on the
  // device, the longest at most n_max), whole scene in the kernel (scenes without deep trees): one launch
  void (*shadow_rays)(hipStream_t, const rptdev::Scene&, const rptdev::PathState&, const uint32_t* sq_all,
                      const uint32_t* sq_counts, uint32_t n_max, int num_lights, double* srt);
  void (*resolve)(hipStream_t, const rptdev::Frame&, const rptdev::PathState&, uint32_t n_samples);
  // means into the full frame, or `packed` into a compact [npix][3] array in the order of Frame::pixels
  void (*finish)(hipStream_t, const rptdev::Frame&, double iterations, double ev_scale, void* out, bool f32, bool packed);
  // multi-GPU gather, root side: one rank's packed pixels into their places in the full f32 frame
  void (*scatter_f32)(hipStream_t, const float* src, const uint32_t* pixels, uint32_t n, float* dst);
  void (*eval_math)(hipStream_t, int fn, uint64_t n, const double* x, const double* y, double* out);
  // persistent per-pixel kernel: resident 64-thread blocks per CU, and the launch
  // (lds_bytes: the flat scene's tables; park: RPT_PATHS_PARK_LDS more behind them for parked environment lookups)
  int (*paths_max_blocks_per_cu)(const FlatLayout* flat /* null: not a flat scene */, uint32_t lds_bytes, bool park);
  void (*paths)(hipStream_t, const rptdev::Scene&, const rptdev::Frame&, const rptdev::Camera&,
                uint32_t* work_counter, double* rec, unsigned long long* ray_counters, double* lbuf, uint32_t spp,
                uint32_t chunk, uint32_t n_items, uint32_t nblocks, const FlatLayout& lay, bool flat, uint32_t lds_bytes,
                bool park, uint32_t batch /* work items a wave claims at a time; 0 = by the launch's size */);
  // pixel sums of a launch's samples, in sample order
  void (*sum_samples)(hipStream_t, const rptdev::Frame&, const double* lbuf, uint32_t spp, bool first);
  // deep-tree scenes: one closest-hit (light < 0) or visibility (light >= 0) query of a depth, run
  // object by object with per-tree ray compaction and persistent traversal
  void (*query)(hipStream_t, const rptdev::Scene&, const rptdev::PathState&, const uint32_t* queue, uint32_t n,
                int light, double* srt, const uint32_t* n_dev /* light >= 0: the device-side length of `queue` */, const uint8_t* obj_deep, const uint8_t* obj_tris, int num_objects,
                uint32_t* tq, uint32_t* tq_ctr, uint32_t trace_blocks, const SortBufs* sort, const QueryHook* hook,
                const StackSpill* spill /* the traversal stack beyond the LDS levels */, QueryTuning* qt);
  size_t (*sort_temp_bytes)(uint32_t n);
  void (*shadow_sum)(hipStream_t, const rptdev::Scene&, const rptdev::PathState&, const uint32_t* queue, uint32_t n,
                     uint32_t rec_off, const double* srt);
  // device-resident Buffer (buffer.rs)
  void (*buffer_add)(hipStream_t, double* total, const double* batch, uint64_t n);
  void (*buffer_image)(hipStream_t, const double* total, uint32_t w, uint32_t h, uint32_t radius, uint32_t nb,
                       const double* thr, uint8_t* out);
  void (*buffer_variance)(hipStream_t, const double* total, const double* const* batches, uint32_t nb, uint64_t npix,
                          double* out);
  // -DRPT_PROF builds: the per-phase table of kernels/prof.inc since the last call ([0] wave cycles, [1] lane cycles,
  // [2] wave iterations, [3] lane iterations); false in regular builds
  bool (*read_prof)(unsigned long long out[4][24]);
  // in-kernel-traversal scenes: the next depth's paths sorted by ray key into the current state arrays (kernels/wavefront.inc)
  void (*path_reorder)(hipStream_t, const rptdev::PathState&, uint32_t n, bool sorted, const SortBufs* sort, uint32_t* order);
};

namespace rpt_strict { extern const KernelTable TABLE; } // -ffp-contract=off (parity mode)
// the same with the extended shape set (RPT_SHAPE_MONOMIAL, trees of trees) compiled in
namespace rpt_strict_ext { extern const KernelTable TABLE; }
