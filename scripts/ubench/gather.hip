// Micro-benchmark: per-CU throughput of per-lane 16-byte gathers (global_load_dwordx4) from an L2-resident table,
// as a function of how many lanes share a 128-byte line.  Answers: what does a divergent gather cost the vector
// memory path (TA/TCP), per lane request — the currency of the kd traversal kernels.
//   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) V4 { unsigned x, y, z, w; };

// every lane walks a pseudo-random sequence of 16-B chunks; lanes are grouped `share` to a 128-B line:
// lane l reads chunk (l % share) of line idx(l / share, step)
template <int SHARE>
__global__ void __launch_bounds__(256) gather(const V4* __restrict__ tab, unsigned n_lines, int steps, unsigned* out, int chain) {
  unsigned lane = threadIdx.x & 63, grp = (blockIdx.x * blockDim.x + threadIdx.x) / SHARE;
  unsigned sub = lane % SHARE;
  unsigned state = grp * 2654435761u + 12345u;
  unsigned acc = 0;
  for (int s = 0; s < steps; s++) {
    V4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { // four independent requests in flight per lane
      state = state * 1664525u + 1013904223u;
      unsigned line = (state >> 8) % n_lines;
      v[k] = tab[(size_t)line * 8 + (sub & 7)];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) acc += v[k].x + v[k].w;
    if (chain) state ^= acc & 1u; // dependent chain variant
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int SHARE> void run(const V4* tab, unsigned n_lines, unsigned* out, int blocks, int steps, const char* what) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(gather<SHARE>, dim3(blocks), dim3(256), 0, 0, tab, n_lines, 8, out, 0);
  hipEventRecord(a);
  hipLaunchKernelGGL(gather<SHARE>, dim3(blocks), dim3(256), 0, 0, tab, n_lines, steps, out, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  double req = (double)blocks * 256 * steps * 4;
  printf("%-34s table %6.1f MB: %8.3f ms  %7.2f G lane-requests/s = %5.2f per CU-cycle @2.4GHz, %7.1f GB/s useful, %6.2f G lines/s\n", what,
         n_lines * 128.0 / 1e6, ms, req / ms / 1e6, req / ms / 1e6 / (256 * 2.4), req * 16 / ms / 1e6, req / SHARE / ms / 1e6);
}

int main() {
  int blocks = 256 * 8 * 4, steps = 256;
  for (unsigned mb : {2u, 24u, 512u}) {
    unsigned n_lines = mb * 1024 * 1024 / 128;
    V4* tab; unsigned* out;
    hipMalloc(&tab, (size_t)n_lines * 128);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMemset(tab, 1, (size_t)n_lines * 128);
    run<1>(tab, n_lines, out, blocks, steps, "1 lane per line (fully divergent)");
    run<2>(tab, n_lines, out, blocks, steps, "2 lanes share a 128-B line");
    run<4>(tab, n_lines, out, blocks, steps, "4 lanes share a 128-B line");
    run<8>(tab, n_lines, out, blocks, steps, "8 lanes share a 128-B line");
    hipFree(tab); hipFree(out);
  }
  return 0;
}
