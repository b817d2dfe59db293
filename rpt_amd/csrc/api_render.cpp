// api_render.cpp — workspace, the wavefront loop and the persistent launch behind every render entry point;
// rptgpu_render_batch[_device], rptgpu_closest_hit, rptgpu_eval_math (see api_internal.h)
#include "api_internal.h"

namespace rptapi {

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

// cap: path slots; rec_cols: columns of the depth-record pool (PathState::rec: one per path and depth REACHED)
void ensure_workspace(rptgpu_scene* h, uint64_t cap, uint64_t rec_cols) {
  if (cap <= h->ws_cap && rec_cols <= h->ws_rec_cols && h->ray.p) return;
  cap = std::max(cap, h->ws_cap);
  rec_cols = std::max(rec_cols, h->ws_rec_cols);
  int nl = std::max(1, h->dscene.num_lights);
  h->ray.alloc(6 * cap);
  h->hit.alloc(4 * cap);
  h->hit_obj.alloc(cap);
  h->draw.alloc(cap); h->pid.alloc(cap); h->col.alloc(cap);
  if (h->path_reorder) { // the survivors' next state as 64-byte rows (gathered in sorted order by rpt_path_permute)
    h->next_rows.alloc(8 * cap);
  } else {               // ... or as a second set of the arrays, swapped per depth
    h->ray_next.alloc(6 * cap);
    h->draw_next.alloc(cap); h->pid_next.alloc(cap); h->col_next.alloc(cap);
  }
  h->last_col.alloc(cap);
  h->rec.release();
  h->rec.alloc((uint64_t)rptdev::REC_FIELDS * rec_cols);
  h->rec_parent.alloc(rec_cols);
  h->shadow.release();
  h->shadow.alloc((uint64_t)nl * rptdev::SHADOW_FIELDS * cap);
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
  if (h->path_reorder) { // the per-depth re-order of the paths (in-kernel-traversal scenes): keys, positions, rocPRIM's scratch
    h->sort_kin.alloc(cap); h->sort_kout.alloc(cap); h->sort_vin.alloc(cap); h->path_order.alloc(cap);
    size_t bytes = rpt_strict::TABLE.sort_temp_bytes((uint32_t)cap);
    h->sort_tmp.alloc(bytes);
    h->sort_bufs = SortBufs{h->sort_kin.p, h->sort_kout.p, h->sort_vin.p, h->sort_tmp.p, bytes};
  }
  h->ws_cap = cap;
  h->ws_rec_cols = rec_cols;
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
  h->ray.release(); h->ray_next.release(); h->hit.release(); h->hit_obj.release(); h->draw.release(); h->draw_next.release();
  h->pid.release(); h->pid_next.release(); h->col.release(); h->col_next.release(); h->rec.release();
  h->rec_parent.release(); h->last_col.release();
  h->shadow.release(); h->tq.release(); h->srt.release(); h->shadow_q.release();
  h->sort_kin.release(); h->sort_kout.release(); h->sort_vin.release(); h->sort_tmp.release(); h->tree_rays.release();
  h->path_order.release(); h->next_rows.release();
  h->gen_defer.release(); h->gen_frame.release(); h->gen_threads = 0; // rpt_tree_generic's columns (ensure_generic makes them again)
  h->ws_cap = 0; h->ws_rec_cols = 0;
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

// packed (with d_out, f32 or f64): d_out receives only this part's pixels, [npix][3] in the order of the part's pixel list
int render_impl(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* p, void* d_out, bool out_f32,
                double* host_out, hipStream_t user_stream, bool packed) {
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
        if (std::getenv("RPTGPU_PRINT_LAUNCH")) { // diagnostics: where the 32-bit work counter ended (kernels/paths.inc fetch_item)
          uint32_t ended = 0;
          HIP_TRY(hipMemcpyAsync(&ended, h->counters.p, sizeof ended, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
          const uint64_t items = (uint64_t)npix * ((spp + chunk - 1) / chunk);
          const uint32_t batch_used = h->opt.paths_batch ? std::min<uint32_t>(h->opt.paths_batch, RPT_PATHS_BATCH_MAX)
                                                          : std::min(256u, std::max((uint32_t)items / (std::max(1u, nblocks) * 32u), 16u));
          std::fprintf(stderr, "rpt_paths work counter: ended at %u for %llu items; %u waves, claims of at most %u: dead claims %lld of at most %llu\n",
                       ended, (unsigned long long)items, nblocks, batch_used, (long long)ended - (long long)items,
                       (unsigned long long)nblocks * (64u + batch_used));
        }
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
      // 106, 128 Mi -> 128 Msamples/s; 16k-triangle glass 179 -> 324; round 6, with passes of 85 % of the free memory:
      // 786 -> 913, profiles/r06_pass_size_ab.txt).  288 GB of HBM is what makes that affordable.
      // What a path costs: its slot (ray, hit, queues, per-light shadow state, the per-tree query's row and sort words) and
      // one 68-byte COLUMN per depth it reaches (PathState::rec) — as many columns as the depths' queues were long, not
      // (max_bounces + 1) per path: the glass's paths average a quarter of their 17 levels.  How many columns a path needs
      // is measured (h->rec_ratio: the first pass of a handle is one sample per pixel with the full pool) and carried
      // with a margin; a pass whose pool runs out at some depth is started over with fewer paths — a pass changes
      // nothing outside the workspace before its rpt_resolve.
      const uint64_t nl = (uint64_t)std::max(1, h->dscene.num_lights);
      // (ray and next ray, hit, object, draw / path id / parent column twice each, last column, per light the shadow
      // state, queue entry and record time, the per-tree query's queue, row and sort words)
      const uint64_t per_slot = 2 * 6 * 8 + 4 * 8 + 4 + 6 * 4 + 4 + nl * rptdev::SHADOW_FIELDS * 8 + nl * (8 + 4) +
                                (h->has_deep ? 12 + 64 + (h->sort_rays ? 12 + 16 : 0) : 0) + (h->path_reorder ? 16 + 16 + 4 : 0);
      const uint64_t per_rec = rptdev::REC_FIELDS * 8 + 4;
      const double full_ratio = (double)p->max_bounces + 1.0;
      if (h->rec_ratio_bounces != p->max_bounces) {
        h->rec_ratio = 0.0; h->rec_ratio_bounces = p->max_bounces;
        // tests: start from a given (too small) figure instead of measuring, so that passes run out of columns and start over
        if (const char* e = std::getenv("RPTGPU_REC_RATIO")) h->rec_ratio = std::max(0.0, std::atof(e));
      }
      const bool generic_all = h->has_deep && (h->gen_all || h->dscene.force_general);
      h->accum.alloc((uint64_t)npix * 3);
      HIP_TRY(hipMemsetAsync(h->accum.p, 0, (uint64_t)npix * 3 * sizeof(double), st));

      rptdev::Frame fr{};
      fr.width = p->width; fr.height = p->height; fr.npix = npix; fr.pixels = h->pixels.p;
      fr.max_bounces = p->max_bounces; fr.seed = p->seed; fr.accum = h->accum.p;
      rptdev::Camera cam = make_camera(*camera);
      const bool any_lights = h->dscene.num_lights > 0;
      const uint32_t nctr = 2u + (uint32_t)h->dscene.num_lights;
      QueryMarks qm(h, prof);
      const QueryHook qhook{query_mark, &qm};

      uint32_t s0 = 0;
      while (s0 < p->iterations) {
        const uint32_t remaining = p->iterations - s0;
        // columns per path of this pass: measured + 10 % + 0.05, the full (max_bounces + 1) until there is a measurement
        // (rounded up to a twentieth, so that the fourth digit of a pass's average does not resize a 100 GB workspace)
        const double ratio = h->rec_ratio > 0.0 ? std::min(full_ratio, std::ceil((h->rec_ratio * 1.10 + 0.05) * 20.0) / 20.0) : full_ratio;
        const double per_path = (double)per_slot + ratio * (double)per_rec;
        uint64_t target = h->target_paths;
        if (!target) {
          uint64_t budget = h->ws_budget_bytes;
          size_t free_b = 0, total_b = 0;
          if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t have = h->ws_cap * per_slot + h->ws_rec_cols * per_rec; // what this handle already holds counts as available
            // the share of the free memory a pass may take (round 6: 85 %, was 1/2 — the passes of a 288 GB device were
            // sized for 140 GB).  RPTGPU_WS_FREE_FRACTION (percent): experiments only
            uint64_t pct = RPT_WS_FREE_PERCENT;
            if (const char* e = std::getenv("RPTGPU_WS_FREE_FRACTION")) pct = (uint64_t)std::min(95, std::max(5, std::atoi(e)));
            budget = std::min<uint64_t>(budget, (free_b + have) / 100 * pct);
          }
          target = std::min<uint64_t>(RPT_MAX_PATHS_PER_PASS, std::max<uint64_t>(1ull << 20, (uint64_t)((double)budget / per_path)));
        }
        // a size that did not fit before is not tried again (several handles or processes on one GPU see the same `free`
        // figure; an explicit target_paths may be more than the device holds): allocating and freeing 100+ GB per call
        // costs seconds
        if (h->ws_fail_paths) target = std::min<uint64_t>(target, h->ws_fail_paths / 2);
        uint32_t s_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(remaining, target / npix));
        if (h->rec_ratio == 0.0 && remaining > 1u && s_chunk > 1u) {
          s_chunk = 1u; // the measuring pass: one sample per pixel, every path with room for all its levels
        } else if (s_chunk < remaining) {
          // passes of EQUAL size: 256 spp with room for 123 per pass are three passes of 86 / 85 / 85, not 123 / 123 / 10
          // (the deep bounces of a 10-spp pass run on a tenth of the rays)
          const uint32_t n_pass = (remaining + s_chunk - 1) / s_chunk;
          s_chunk = (remaining + n_pass - 1) / n_pass;
        }
        // several handles (or processes) on one GPU each see the same `free` figure: if the pass does not fit after
        // all, halve it instead of failing the render (a smaller pass is only slower)
        // (rpt_tree_generic's large grid — whole objects, or under RPT_FLAG_GENERAL_TRAVERSAL everything, go through it: up
        // to several hundred MB of columns for a deep mesh — is part of the same attempt: if it does not fit, the pass shrinks)
        // The workspace is made for the pass a call of this size runs once the measurement is in — not for this pass's
        // own size: the first call's passes are 1 + (n - 1) samples, and a workspace of n - 1 would be freed and made
        // again by the second call (220 GB: six seconds)
        uint32_t s_alloc = s_chunk;
        if (h->rec_ratio > 0.0) s_alloc = (uint32_t)std::max<uint64_t>(s_chunk, std::min<uint64_t>(p->iterations, target / npix));
        uint64_t rec_cols = 0;
        for (;;) {
          const uint64_t np = (uint64_t)npix * std::max(s_chunk, s_alloc);
          rec_cols = std::max<uint64_t>(np, std::min<uint64_t>((uint64_t)std::ceil((double)np * ratio), 0xfffffff0ull));
          try {
            ensure_workspace(h, np, rec_cols);
            if (generic_all) ensure_generic(h, true);
            break;
          } catch (const HipError& e) {
            if (e.e != hipErrorOutOfMemory || s_chunk == 1) throw;
            (void)hipGetLastError(); // clear the sticky error before retrying
            release_workspace(h);
            h->ws_fail_paths = h->ws_fail_paths ? std::min<uint64_t>(h->ws_fail_paths, np) : np;
            if (s_alloc > s_chunk) s_alloc = s_chunk; // (first the room ahead goes, then the pass shrinks)
            else { s_chunk = std::max(1u, s_chunk / 2); s_alloc = s_chunk; }
          }
        }
        rec_cols = h->ws_rec_cols; // (a workspace kept from an earlier, larger pass: all of its columns)

        rptdev::PathState ps{};
        ps.ray = h->ray.p; ps.hit = h->hit.p; ps.hit_obj = h->hit_obj.p; ps.draw = h->draw.p;
        ps.pid = h->pid.p; ps.col = h->col.p;
        ps.ray_next = h->ray_next.p; ps.draw_next = h->draw_next.p; ps.pid_next = h->pid_next.p; ps.col_next = h->col_next.p;
        ps.rec = h->rec.p; ps.rec_parent = h->rec_parent.p; ps.last_col = h->last_col.p;
        ps.shadow = h->shadow.p; ps.cap = h->ws_cap; ps.rec_cap = h->ws_rec_cols;
        if (h->path_reorder) { // (never with per-tree queues: api_scene.cpp)
          ps.sort_keys = h->sort_kin.p; ps.sort_vals = h->sort_vin.p; ps.next_rows = h->next_rows.p;
          std::memcpy(ps.key_bounds, h->scene_bounds, sizeof ps.key_bounds);
        }
        // the counter sets the kernels clear for each other start cleared (one memset per pass, not one per depth and
        // per tree and query: 102 of the wine glass's 354 fills per step)
        HIP_TRY(hipMemsetAsync(h->counters.p, 0, 2 * (size_t)nctr * sizeof(uint32_t), st));
        uint32_t cset = 0;
        if (h->has_deep) {
          HIP_TRY(hipMemsetAsync(h->tq_ctr.p, 0, 16 * sizeof(uint32_t), st));
          h->qtune.ctr_set = 0;
        }
        const RptStats stats_at_start = h->stats; // (a pass that is started over counts once)

        const uint32_t sc = s_chunk;
        const uint32_t n_paths = npix * sc;
        fr.sample_base = p->sample_index_base + s0;
        { Bracket b(h, RPT_K_RAYGEN, prof); kt->raygen(st, fr, cam, ps, n_paths); b.done(); }
        h->stats.samples += n_paths;
        uint32_t n_active = n_paths;
        const uint32_t* const queue = nullptr; // the paths of a depth stand densely in its state arrays: the identity
        uint32_t* const next = nullptr;
        uint64_t rec_off = 0; // the depth's first record column
        bool pool_ran_out = false;
        for (uint32_t depth = 0; depth <= p->max_bounces && n_active; depth++) {
          if (rec_off + n_active > rec_cols) { pool_ran_out = true; break; }
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
          const int nlights = h->dscene.num_lights;
          uint32_t* const ctrs = h->counters.p + (size_t)cset * nctr;       // this depth's counters (cleared by the depth before)
          uint32_t* const ctrs_next = h->counters.p + (size_t)(cset ^ 1u) * nctr;
          cset ^= 1u;
          { Bracket b(h, RPT_K_SHADE, prof);
            kt->shade(st, h->dscene, fr, ps, queue, n_active, depth, next, ctrs, h->shadow_q.p, ctrs_next, nctr, (uint32_t)rec_off); b.done(); }
          // The depth's counts come back right after rpt_shade — the one point of a depth where the host waits — so the
          // visibility queries are sized for the shadow rays there ARE (50-70 % of the paths on closed meshes: less to
          // sort, smaller grids, and a light without a single ray at this depth costs no launch at all) and the next
          // depth for its survivors.  Until round 5 the wait stood at the depth's end and the queries ran over the
          // host's bound, the number of paths.  Everything up to the next rpt_shade is then enqueued without a wait.
          h->cnt_host.resize(2 + (size_t)nlights);
          uint32_t* cnt = h->cnt_host.data();
          HIP_TRY(hipMemcpyAsync(cnt, ctrs, (2 + (size_t)nlights) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
          for (int l = 0; l < nlights; l++) h->stats.shadow_rays_traced += cnt[2 + l];
          if (prof && h->pending.size() >= 256) drain_events(h); // the stream is idle here: cheap
          h->stats.shadow_rays += (uint64_t)cnt[1] * (uint64_t)h->dscene.num_shadow_lights;
          if (any_lights) {
            // the visibility queries run over rpt_shade's per-light shadow-ray queues (their lengths also stay on the
            // device: ctrs + 2 + l is what the kernels read)
            Bracket b(h, RPT_K_SHADOW, prof);
            if (by_object) {
              for (int l = 0; l < nlights; l++)
                if (h->light_casts[l] && cnt[2 + l])
                  kt->query(st, h->dscene, ps, h->shadow_q.p + (uint64_t)l * ps.cap, cnt[2 + l], l, h->srt.p, ctrs + 2 + l, h->obj_deep.data(), h->obj_tris.data(),
                            h->dscene.num_objects, h->tq.p, h->tq_ctr.p, trace_blocks, h->sort_rays ? &h->sort_bufs : nullptr, &qhook, &h->spill, &h->qtune);
            } else { // one launch for all lights of the depth (the grid's y is the light)
              uint32_t n_max = 0;
              for (int l = 0; l < nlights; l++)
                if (h->light_casts[l]) n_max = std::max(n_max, cnt[2 + l]);
              if (n_max) kt->shadow_rays(st, h->dscene, ps, h->shadow_q.p, ctrs + 2, n_max, nlights, h->srt.p);
            }
            kt->shadow_sum(st, h->dscene, ps, queue, n_active, (uint32_t)rec_off, h->srt.p);
            b.done();
          }
          rec_off += n_active;
          n_active = cnt[0];
          // the survivors' state is what rpt_shade wrote to the *_next arrays at their new positions
          if (h->path_reorder) {
            // ... as rows, gathered into the current arrays in the order of their rays' keys — or, a depth too small to be
            // worth a sort, as they stand (behind the depth's shadow queries, which read the current arrays: same stream)
            if (n_active && depth < p->max_bounces) {
              Bracket b(h, RPT_K_TREE_SORT, prof);
              kt->path_reorder(st, ps, n_active, n_active >= h->path_reorder_min, &h->sort_bufs, h->path_order.p);
              b.done();
            }
          } else {
            std::swap(ps.ray, ps.ray_next); std::swap(ps.draw, ps.draw_next); std::swap(ps.pid, ps.pid_next); std::swap(ps.col, ps.col_next);
          }
        }
        if (pool_ran_out) {
          // more levels per path than the pool was sized for (another camera, a margin too thin): the pass starts over
          // with room for half as many paths again per column budget; nothing of it has left the workspace
          HIP_TRY(hipStreamSynchronize(st));
          h->stats = stats_at_start;
          const double seen = (double)(rec_off + n_active) / (double)n_paths; // a lower bound of what it needs
          h->rec_ratio = std::min(full_ratio, std::max(h->rec_ratio, seen) * 1.5);
          if (std::getenv("RPTGPU_PRINT_LAUNCH"))
            std::fprintf(stderr, "wavefront pass of %u spp started over: the record pool (%.2f columns per path) ran out; now %.2f\n", sc, ratio, h->rec_ratio);
          continue;
        }
        { Bracket b(h, RPT_K_RESOLVE, prof); kt->resolve(st, fr, ps, sc); b.done(); }
        HIP_TRY(hipGetLastError()); // a failed launch is reported here, not by the stream sync
        h->rec_ratio = std::max(h->rec_ratio, (double)rec_off / (double)n_paths); // columns used per path: the largest average seen
        if (std::getenv("RPTGPU_PRINT_LAUNCH"))
          std::fprintf(stderr, "wavefront pass: %u spp, %u paths, %llu record columns used of %llu (%.3f per path, pool sized for %.3f)\n",
                       sc, n_paths, (unsigned long long)rec_off, (unsigned long long)rec_cols, (double)rec_off / (double)n_paths, ratio);
        s0 += sc;
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

} // namespace rptapi

extern "C" {

int rptgpu_render_batch(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params, double* out_rgb) {
  if (!out_rgb) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null out_rgb");
  return render_impl(h, camera, params, nullptr, false, out_rgb, nullptr);
}

int rptgpu_render_batch_device(rptgpu_scene* h, const RptCamera* camera, const RptRenderParams* params, void* d_out,
                               int out_is_f32, void* stream) {
  if (!d_out) return fail(h, RPTGPU_E_INVALID_ARGUMENT, "null d_out");
  return render_impl(h, camera, params, d_out, out_is_f32 != 0, nullptr, (hipStream_t)stream);
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
      ensure_workspace(h, piece, piece);
      rptdev::PathState ps{};
      ps.ray = h->ray.p; ps.hit = h->hit.p; ps.hit_obj = h->hit_obj.p; ps.draw = h->draw.p;
      ps.rec = h->rec.p; ps.rec_parent = h->rec_parent.p; ps.last_col = h->last_col.p;
      ps.shadow = h->shadow.p; ps.cap = h->ws_cap; ps.rec_cap = h->ws_rec_cols;
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

} // extern "C"
