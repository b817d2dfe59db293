// kernels.h — host-callable launchers of the gfx950 kernels, one table per arithmetic mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

struct KernelTable {
  void (*raygen)(hipStream_t, const rptdev::Frame&, const rptdev::Camera&, const rptdev::PathState&, uint32_t n_paths);
  void (*extend)(hipStream_t, const rptdev::Scene&, const rptdev::PathState&, const uint32_t* queue, uint32_t n);
  void (*extend_rays)(hipStream_t, const rptdev::Scene&, const double* o, const double* d, uint64_t n, double* out_t,
                      double* out_n, int32_t* out_obj);
  void (*shade)(hipStream_t, const rptdev::Scene&, const rptdev::Frame&, const rptdev::PathState&,
                const uint32_t* queue, uint32_t n, uint32_t depth, uint32_t* next_queue, uint32_t* counters);
  void (*shadow)(hipStream_t, const rptdev::Scene&, const rptdev::PathState&, const uint32_t* queue, uint32_t n,
                 uint32_t depth);
  void (*resolve)(hipStream_t, const rptdev::Frame&, const rptdev::PathState&, uint32_t n_samples);
  void (*finish)(hipStream_t, const rptdev::Frame&, double iterations, double ev_scale, void* out, bool f32);
  void (*eval_math)(hipStream_t, int fn, uint64_t n, const double* x, const double* y, double* out);
  // persistent per-pixel kernel: resident 64-thread blocks per CU, and the launch
  int (*paths_max_blocks_per_cu)();
  void (*paths)(hipStream_t, const rptdev::Scene&, const rptdev::Frame&, const rptdev::Camera&,
                uint32_t* work_counter, double* rec, unsigned long long* ray_counters, uint32_t spp,
                uint32_t nblocks);
};

namespace rpt_strict { extern const KernelTable TABLE; } // -ffp-contract=off (parity mode)
namespace rpt_fast { extern const KernelTable TABLE; }   // -ffp-contract=fast
