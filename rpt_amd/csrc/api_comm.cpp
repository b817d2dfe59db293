// api_comm.cpp — the multi-GPU side of the ABI: communicator, rptgpu_render_batch_reduce (the library-owned exchange over RCCL
// and its failure paths), rptgpu_render_batch_emulate_ranks (see api_internal.h)
#include "api_internal.h"

extern "C" {

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

} // extern "C"
