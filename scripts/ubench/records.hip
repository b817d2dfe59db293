// Micro-benchmark for the leaf phase of the kd traversal: every lane needs ONE 128-byte record (a triangle's
// intersection record) at a random position of a table.
//   A  each lane loads its own record: 8 global_load_dwordx4 per lane, 64 distinct lines per wave instruction
//   B  the wave loads the same 64 records cooperatively: instruction k covers records 8k..8k+7, lane l reads chunk
//      l%8 of record 8k + l/8 (8 lanes = one 128-B line), stores it to LDS, then every lane reads its own record
//      from LDS (8 ds_read_b128)
//   C  like A but only the first 48 bytes (3 chunks) per lane;  D  like B for 48 bytes (3 instructions, 21 records each)
// Reported: records per second and per CU-cycle.
//   hipcc --offload-arch=gfx950 -O3 records.hip -o records && ./records
#include <hip/hip_runtime.h>
#include <cstdio>

struct alignas(16) V4 { unsigned x, y, z, w; };
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const V4* __restrict__ tab, unsigned mask, int steps, unsigned* out) {
  __shared__ V4 buf[4][64 * 8];
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (int s = 0; s < steps; s++) {
    const unsigned rec = hash(gid * 9781u + s * 6271u) & mask; // this lane's record
    if (MODE == 0 || MODE == 2) {
      constexpr int N = MODE == 0 ? 8 : 3;
      V4 v[N];
#pragma unroll
      for (int c = 0; c < N; c++) v[c] = tab[(size_t)rec * 8 + c];
#pragma unroll
      for (int c = 0; c < N; c++) acc += v[c].x ^ v[c].w;
    } else if (MODE == 1) {
      V4 v[8];
#pragma unroll
      for (int kk = 0; kk < 8; kk++) { // records 8kk .. 8kk+7 of the wave, chunk lane%8
        unsigned r = __shfl(rec, kk * 8 + (lane >> 3));
        v[kk] = tab[(size_t)r * 8 + (lane & 7)];
      }
#pragma unroll
      for (int kk = 0; kk < 8; kk++) buf[wv][(kk * 8 + (lane >> 3)) * 8 + (lane & 7)] = v[kk];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < 8; c++) { V4 t = buf[wv][lane * 8 + c]; acc += t.x ^ t.w; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else { // MODE 3: 48 bytes cooperatively: 192 chunks = 3 instructions
      V4 v[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        unsigned ch = kk * 64 + lane, owner = ch / 3, c = ch - owner * 3;
        unsigned r = __shfl(rec, owner);
        v[kk] = tab[(size_t)r * 8 + c];
      }
#pragma unroll
      for (int kk = 0; kk < 3; kk++) buf[wv][kk * 64 + lane] = v[kk];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < 3; c++) { V4 t = buf[wv][lane * 3 + c]; acc += t.x ^ t.w; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  out[gid] = acc;
}

template <int MODE> void run(const V4* tab, unsigned n_rec, unsigned* out, const char* what) {
  const int blocks = 256 * 4 * 2, steps = 512;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, n_rec - 1, 8, out);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, n_rec - 1, steps, out);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  double recs = (double)blocks * 256 * steps;
  printf("%-52s table %7.2f MB: %8.3f ms  %7.2f G records/s = %5.3f per CU-cycle @2.4GHz\n", what, n_rec * 128.0 / 1e6, ms,
         recs / ms / 1e6, recs / ms / 1e6 / (256 * 2.4));
}

int main() {
  for (unsigned lg : {7u, 14u, 19u, 23u}) { // 16 KB (L1), 2 MB (L2), 64 MB (the mesh's leaf records), 1 GB (HBM)
    unsigned n_rec = 1u << lg;
    V4* tab; unsigned* out;
    (void)hipMalloc(&tab, (size_t)n_rec * 128);
    (void)hipMalloc(&out, (size_t)256 * 4 * 2 * 256 * 4);
    (void)hipMemset(tab, 1, (size_t)n_rec * 128);
    run<0>(tab, n_rec, out, "A per-lane record, 128 B (8 dwordx4)");
    run<1>(tab, n_rec, out, "B cooperative 128 B via LDS (8 lanes per record)");
    run<2>(tab, n_rec, out, "C per-lane first 48 B (3 dwordx4)");
    run<3>(tab, n_rec, out, "D cooperative 48 B via LDS (3 lanes per record)");
    (void)hipFree(tab); (void)hipFree(out);
  }
  return 0;
}
