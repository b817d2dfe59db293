// kdbuild.hip — KdTree::new / construct (src/kdtree.rs:108-119, 235-355) on the device, node for node the tree
// host_scene.cpp's kd_build makes (and therefore the reference's).
//
// The reference sorts the 2n box edges of a node on every axis at every node (kdtree.rs:252-255) to find three medians.
// Here the 2n edges ("events") are sorted ONCE per axis for the whole primitive set (rocPRIM radix sort, f64 keys);
// a split distributes a node's primitives to its children with a STABLE partition, so a child's events stay in
// sorted order and its medians are the two middle events of its segment — no further sorting.  The build is
// level-synchronous: one round of kernels per tree level over ALL nodes of that level (a few scans and scatters over
// the level's primitive instances, whatever the number of nodes), ~20 launches and one small read-back per level.
//
//   per level:  medians + extents (two middle events, first / last event)            one thread per node
//               partition scores l = #{lo <= m}, r = #{hi >= m} per axis              one thread per instance
//               the decision of construct(): leaf, or (axis, value), child sizes      one thread per node
//               stable scatter of the instances and of the three event lists          scans + one thread per item
//
// The host then renumbers the breadth-first result in the depth-first order of construct() (children allocated as a
// pair when their parent is split, left subtree first) — the numbering every fixture and the device traversal use.
// Straddling primitives go to both children (kdtree.rs:270-281), so a level holds more instances than primitives; the
// buffers grow on demand.  Compiled with -ffp-contract=off: the medians are (a + b) / 2 in IEEE f64 as on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "host_scene.h"

namespace rpthost {

namespace {

struct HipErr {
  hipError_t e;
  const char* what;
  int line;
};
#define KD_TRY(expr)                                          \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) throw HipErr{_e, #expr, __LINE__};  \
  } while (0)

constexpr double SCORE_THRESHOLD = 0.85; // kdtree.rs:6
constexpr uint32_t NONE = 0xFFFFFFFFu;

// one node of the level being built
struct Task {
  uint32_t start, count; // its primitive instances: inst[start, start + count); its events: ev[k][2 start, 2 start + 2 count)
  double clo[3], chi[3]; // the node's cell (root bounds cut by the ancestors' split planes): the "regular" check
};
struct Decision {
  double m[3];        // medians per axis
  double ext[3];      // extent of the primitives' boxes per axis (kdtree.rs:291-293)
  uint32_t l[3], r[3]; // partition scores' operands (kdtree.rs:257-268)
  int32_t dir;        // -1: leaf
  double value;
  uint32_t split_rank; // index among the level's split nodes (children: tasks 2 rank, 2 rank + 1 of the next level)
};
// breadth-first node record handed to the host
struct BfsNode {
  double split;
  uint32_t a;    // inner: index of the left child within the NEXT level; leaf: first entry in the leaf buffer
  uint32_t info; // inner: axis (0..2); leaf: 3 | count << 2
};

struct Boxes { // structure of arrays on the device
  const double* lo[3];
  const double* hi[3];
};

__device__ __forceinline__ double ev_value(const Boxes& b, int k, uint32_t prim, uint32_t is_hi) {
  return is_hi ? b.hi[k][prim] : b.lo[k][prim];
}

// ---- level 0: events ------------------------------------------------------------------------------------------
__global__ void k_make_events(Boxes b, int k, uint32_t n, double* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // (the KEY of a zero is +0.0 whatever its sign: the reference's comparison holds -0.0 == +0.0 and its sort is
  // stable, kdtree.rs:251-254, so zeros of both signs stay in pushed order — the radix sort would put every -0.0
  // first; the medians are read from the boxes, signs intact)
  const double lo = b.lo[k][i], hi = b.hi[k][i];
  keys[2 * i] = lo == 0.0 ? 0.0 : lo; vals[2 * i] = i << 1;
  keys[2 * i + 1] = hi == 0.0 ? 0.0 : hi; vals[2 * i + 1] = (i << 1) | 1u;
}
__global__ void k_iota(uint32_t n, uint32_t* __restrict__ inst, uint32_t* __restrict__ task_of) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  inst[i] = i;
  task_of[i] = 0;
}

// ---- per level ------------------------------------------------------------------------------------------------
// medians (kdtree.rs:252-255 + median(), :347-355: 2 count values, an even number: the mean of the two middle ones)
// and extents (the smallest event is the smallest lo, the largest the largest hi: kdtree.rs:240-249's fold)
__global__ void k_medians(Boxes b, const Task* __restrict__ tasks, uint32_t nt, const uint32_t* __restrict__ inst,
                          const uint32_t* __restrict__ ev0, const uint32_t* __restrict__ ev1,
                          const uint32_t* __restrict__ ev2, Decision* __restrict__ dec) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const Task tk = tasks[t];
  Decision d;
  for (int k = 0; k < 3; k++) { d.m[k] = 0.0; d.ext[k] = 0.0; d.l[k] = 0; d.r[k] = 0; }
  d.dir = -1; d.value = 0.0; d.split_rank = 0;
  if (tk.count >= 16u) { // kdtree.rs:236-238
    const uint32_t* evs[3] = {ev0, ev1, ev2};
    for (int k = 0; k < 3; k++) {
      const uint32_t* e = evs[k] + 2ull * tk.start;
      auto val = [&](uint32_t pos) { uint32_t x = e[pos]; return ev_value(b, k, inst[tk.start + (x >> 1)], x & 1u); };
      double lo_mid = val(tk.count - 1u), hi_mid = val(tk.count);
      d.m[k] = (hi_mid + lo_mid) / 2.0;
      d.ext[k] = val(2u * tk.count - 1u) - val(0u);
    }
  }
  dec[t] = d;
}

// partition_score's counts (kdtree.rs:257-268) for the three medians, one thread per instance
__global__ void k_scores(Boxes b, const Task* __restrict__ tasks, const uint32_t* __restrict__ inst,
                         const uint32_t* __restrict__ task_of, uint32_t n, Decision* __restrict__ dec) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = j < n;
  uint32_t t = live ? task_of[j] : NONE;
  if (live && tasks[t].count < 16u) live = false;
  uint32_t p = live ? inst[j] : 0u;
  // a wave whose lanes all belong to one node adds its counts with six atomics; mixed waves (small nodes) add per lane
  const uint32_t t0 = __shfl(t, 0);
  const bool uniform = __ballot(t != t0) == 0ull;
  for (int k = 0; k < 3; k++) {
    bool le = false, ge = false;
    if (live) {
      const double m = dec[t].m[k];
      le = b.lo[k][p] <= m;
      ge = b.hi[k][p] >= m;
    }
    if (uniform) {
      const uint64_t ml = __ballot(le), mg = __ballot(ge);
      if ((threadIdx.x & 63u) == 0u && t0 != NONE) {
        if (ml) atomicAdd(&dec[t0].l[k], (uint32_t)__popcll(ml));
        if (mg) atomicAdd(&dec[t0].r[k], (uint32_t)__popcll(mg));
      }
    } else if (live) {
      if (le) atomicAdd(&dec[t].l[k], 1u);
      if (ge) atomicAdd(&dec[t].r[k], 1u);
    }
  }
}

// the decision of construct() (kdtree.rs:286-319); flags for the scans that follow
__global__ void k_decide(const Task* __restrict__ tasks, uint32_t nt, Decision* __restrict__ dec,
                         uint32_t* __restrict__ is_split, uint32_t* __restrict__ leaf_count,
                         uint32_t* __restrict__ child_count /* [2 nt] */, uint32_t* __restrict__ irregular) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == nt) { is_split[nt] = 0u; leaf_count[nt] = 0u; child_count[2 * nt] = 0u; } // the scans' extra element: totals
  if (t >= nt) return;
  const Task tk = tasks[t];
  Decision d = dec[t];
  int dir = -1;
  if (tk.count >= 16u) {
    const unsigned long long n = tk.count;
    unsigned long long s[3];
    for (int k = 0; k < 3; k++) s[k] = d.l[k] > d.r[k] ? d.l[k] : d.r[k];
    const unsigned long long threshold = (unsigned long long)((double)n * SCORE_THRESHOLD); // kdtree.rs:286
    const unsigned long long smin = s[0] < s[1] ? (s[0] < s[2] ? s[0] : s[2]) : (s[1] < s[2] ? s[1] : s[2]);
    if (!(smin >= threshold)) {
      const double ex = d.ext[0], ey = d.ext[1], ez = d.ext[2];
      if (ex > ey && ex > ez) {
        if (s[0] < threshold) dir = 0;
      } else if (ey > ez) {
        if (s[1] < threshold) dir = 1;
      } else if (s[2] < threshold) {
        dir = 2;
      }
      if (dir == -1) {
        if (s[0] < s[1] && s[0] < s[2]) dir = 0;
        else if (s[1] < s[2]) dir = 1;
        else dir = 2;
      }
    }
  }
  d.dir = dir;
  if (dir >= 0) {
    d.value = d.m[dir];
    if (!(d.value >= tk.clo[dir] && d.value <= tk.chi[dir])) atomicOr(irregular, 1u);
    is_split[t] = 1u; leaf_count[t] = 0u;
    child_count[2 * t] = d.l[dir];      // #{lo <= value}: the left list (kdtree.rs:273-275)
    child_count[2 * t + 1] = d.r[dir];  // #{hi >= value}: the right list
  } else {
    is_split[t] = 0u; leaf_count[t] = tk.count;
    child_count[2 * t] = 0u; child_count[2 * t + 1] = 0u;
  }
  dec[t] = d;
}

// node records of this level + the next level's tasks (after the scans over is_split / child_count / leaf_count)
__global__ void k_emit(const Task* __restrict__ tasks, uint32_t nt, Decision* __restrict__ dec,
                       const uint32_t* __restrict__ split_rank, const uint32_t* __restrict__ child_start,
                       const uint32_t* __restrict__ leaf_start, uint32_t leaf_base, BfsNode* __restrict__ nodes,
                       Task* __restrict__ next) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const Task tk = tasks[t];
  Decision d = dec[t];
  BfsNode nd;
  if (d.dir >= 0) {
    const uint32_t r = split_rank[t];
    d.split_rank = r;
    dec[t] = d;
    nd.split = d.value; nd.a = 2u * r; nd.info = (uint32_t)d.dir;
    Task a = tk, b = tk; // BoundingBox::split (kdtree.rs:71-86)
    a.start = child_start[2 * t]; a.count = d.l[d.dir]; a.chi[d.dir] = d.value;
    b.start = child_start[2 * t + 1]; b.count = d.r[d.dir]; b.clo[d.dir] = d.value;
    next[2 * r] = a;
    next[2 * r + 1] = b;
  } else {
    nd.split = 0.0; nd.a = leaf_base + leaf_start[t]; nd.info = 3u | (tk.count << 2);
  }
  nodes[t] = nd;
}

// per instance: does it go left / right (kdtree.rs:273-279)?  packed for ONE scan: low word left, high word right
__global__ void k_flags(Boxes b, const Decision* __restrict__ dec, const uint32_t* __restrict__ inst,
                        const uint32_t* __restrict__ task_of, uint32_t n, unsigned long long* __restrict__ f) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const Decision& d = dec[task_of[j]];
  unsigned long long v = 0ull;
  if (d.dir >= 0) {
    const uint32_t p = inst[j];
    if (b.lo[d.dir][p] <= d.value) v |= 1ull;
    if (b.hi[d.dir][p] >= d.value) v |= 1ull << 32;
  }
  f[j] = v;
}

// stable scatter of the instances into the children's lists (the order of `idx` is kept, kdtree.rs:270-281), leaves
// into the leaf buffer; the instance's positions in the next level are kept for the events
__global__ void k_scatter_inst(const Task* __restrict__ tasks, const Decision* __restrict__ dec,
                               const uint32_t* __restrict__ inst, const uint32_t* __restrict__ task_of, uint32_t n,
                               const unsigned long long* __restrict__ f, const unsigned long long* __restrict__ sf,
                               const uint32_t* __restrict__ child_start, const uint32_t* __restrict__ leaf_start,
                               uint32_t leaf_base, uint32_t* __restrict__ inst_next, uint32_t* __restrict__ task_next,
                               uint32_t* __restrict__ pos_l, uint32_t* __restrict__ pos_r, uint32_t* __restrict__ leaves) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t t = task_of[j];
  const Task& tk = tasks[t];
  const Decision& d = dec[t];
  const uint32_t p = inst[j];
  uint32_t pl = NONE, pr = NONE;
  if (d.dir >= 0) {
    const unsigned long long base = sf[tk.start], mine = sf[j], fl = f[j];
    if (fl & 1ull) {
      pl = child_start[2 * t] + ((uint32_t)mine - (uint32_t)base);
      inst_next[pl] = p; task_next[pl] = 2u * d.split_rank;
    }
    if (fl >> 32) {
      pr = child_start[2 * t + 1] + ((uint32_t)(mine >> 32) - (uint32_t)(base >> 32));
      inst_next[pr] = p; task_next[pr] = 2u * d.split_rank + 1u;
    }
  } else {
    leaves[leaf_base + leaf_start[t] + (j - tk.start)] = p;
  }
  pos_l[j] = pl; pos_r[j] = pr;
}

// the same for one axis' events: an event follows its instance; within a child the sorted order is kept
__global__ void k_ev_flags(const Task* __restrict__ tasks, const uint32_t* __restrict__ task_of,
                           const uint32_t* __restrict__ ev, uint32_t n2, const uint32_t* __restrict__ pos_l,
                           const uint32_t* __restrict__ pos_r, const uint32_t* __restrict__ ev_task,
                           unsigned long long* __restrict__ f) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n2) return;
  const uint32_t t = ev_task[e];
  const uint32_t j = tasks[t].start + (ev[e] >> 1);
  unsigned long long v = 0ull;
  if (pos_l[j] != NONE) v |= 1ull;
  if (pos_r[j] != NONE) v |= 1ull << 32;
  f[e] = v;
}
__global__ void k_ev_scatter(const Task* __restrict__ tasks, const Task* __restrict__ next, const Decision* __restrict__ dec,
                             const uint32_t* __restrict__ ev, uint32_t n2, const uint32_t* __restrict__ pos_l,
                             const uint32_t* __restrict__ pos_r, const uint32_t* __restrict__ ev_task,
                             const unsigned long long* __restrict__ f, const unsigned long long* __restrict__ sf,
                             uint32_t* __restrict__ ev_next, uint32_t* __restrict__ ev_task_next) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n2) return;
  const uint32_t t = ev_task[e];
  const Decision& d = dec[t];
  if (d.dir < 0) return;
  const Task& tk = tasks[t];
  const uint32_t x = ev[e], j = tk.start + (x >> 1);
  const unsigned long long base = sf[2ull * tk.start], mine = sf[e], fl = f[e];
  if (fl & 1ull) {
    const Task& c = next[2u * d.split_rank];
    const uint32_t q = 2u * c.start + ((uint32_t)mine - (uint32_t)base);
    ev_next[q] = ((pos_l[j] - c.start) << 1) | (x & 1u);
    ev_task_next[q] = 2u * d.split_rank;
  }
  if (fl >> 32) {
    const Task& c = next[2u * d.split_rank + 1u];
    const uint32_t q = 2u * c.start + ((uint32_t)(mine >> 32) - (uint32_t)(base >> 32));
    ev_next[q] = ((pos_r[j] - c.start) << 1) | (x & 1u);
    ev_task_next[q] = 2u * d.split_rank + 1u;
  }
}
// the sorted (value, primitive << 1 | is_hi) pairs of level 0 as events of task 0: instance index = primitive index
__global__ void k_ev_task0(uint32_t n2, uint32_t* __restrict__ ev_task) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n2) ev_task[e] = 0u;
}

__global__ void k_totals(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* out) {
  out[0] = *a; out[1] = *b; out[2] = *c;
}

inline dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// one device allocation per build, carved up front: hipMalloc / hipFree cost more than whole levels of this build
struct Arena {
  char* base = nullptr;
  size_t size = 0, off = 0;
  ~Arena() { if (base) (void)hipFree(base); }
  template <class T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += std::max<size_t>(count, 1) * sizeof(T);
    return p;
  }
};

enum class Attempt { done, grow, failed };

// One attempt with room for `cap` primitive instances per level.  Attempt::grow: some level needed more.
Attempt build_once(const std::vector<double>& soa, size_t n, const Box& root, size_t cap, hipStream_t st, uint32_t* pinned,
                   KdBuild& out, std::string& why) {
  const size_t task_cap = cap / 2 + 64, node_cap = cap + 64, leaf_cap = 2 * cap + 64;
  size_t t_sort = 0, t_s32 = 0, t_s64 = 0;
  KD_TRY(rocprim::radix_sort_pairs(nullptr, t_sort, (double*)nullptr, (double*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 2 * n, 0, 64, st));
  KD_TRY(rocprim::exclusive_scan(nullptr, t_s32, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 2 * task_cap + 1, rocprim::plus<uint32_t>(), st));
  KD_TRY(rocprim::exclusive_scan(nullptr, t_s64, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, 2 * cap,
                                 rocprim::plus<unsigned long long>(), st));
  const size_t tmp_bytes = std::max(t_sort, std::max(t_s32, t_s64)) + 256;
  // the layout is walked twice: once over a null arena to measure it, once over the allocation
  double *d_soa = nullptr, *keys_in = nullptr, *keys_out = nullptr;
  uint32_t *inst[2], *task_of[2], *ev[2][3], *ev_task[2], *pos_l = nullptr, *pos_r = nullptr, *leaves = nullptr;
  unsigned long long *fl = nullptr, *sfl = nullptr;
  Task* tasks[2];
  Decision* dec = nullptr;
  uint32_t *is_split = nullptr, *split_rank = nullptr, *leaf_count = nullptr, *leaf_start = nullptr, *child_count = nullptr,
           *child_start = nullptr, *flag = nullptr, *d_totals = nullptr, *vals_in = nullptr;
  BfsNode* d_nodes = nullptr;
  void* tmp = nullptr;
  auto layout = [&](Arena& ar) {
    d_soa = ar.take<double>(6 * n);
    for (int h = 0; h < 2; h++) {
      inst[h] = ar.take<uint32_t>(cap); task_of[h] = ar.take<uint32_t>(cap);
      for (int k = 0; k < 3; k++) ev[h][k] = ar.take<uint32_t>(2 * cap);
      ev_task[h] = ar.take<uint32_t>(2 * cap); // (an event's node does not depend on the axis: the segments coincide)
    }
    pos_l = ar.take<uint32_t>(cap); pos_r = ar.take<uint32_t>(cap); leaves = ar.take<uint32_t>(leaf_cap);
    fl = ar.take<unsigned long long>(2 * cap); sfl = ar.take<unsigned long long>(2 * cap);
    tasks[0] = ar.take<Task>(task_cap); tasks[1] = ar.take<Task>(task_cap);
    dec = ar.take<Decision>(task_cap);
    is_split = ar.take<uint32_t>(task_cap + 1); split_rank = ar.take<uint32_t>(task_cap + 1);
    leaf_count = ar.take<uint32_t>(task_cap + 1); leaf_start = ar.take<uint32_t>(task_cap + 1);
    child_count = ar.take<uint32_t>(2 * task_cap + 1); child_start = ar.take<uint32_t>(2 * task_cap + 1);
    flag = ar.take<uint32_t>(4); d_totals = ar.take<uint32_t>(4);
    d_nodes = ar.take<BfsNode>(node_cap);
    keys_in = ar.take<double>(2 * n); keys_out = ar.take<double>(2 * n); vals_in = ar.take<uint32_t>(2 * n);
    tmp = ar.take<char>(tmp_bytes);
  };
  Arena measure;
  layout(measure);
  Arena ar;
  ar.size = measure.off + 256;
  KD_TRY(hipMalloc((void**)&ar.base, ar.size));
  layout(ar);

  KD_TRY(hipMemcpyAsync(d_soa, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, st));
  Boxes bx;
  for (int k = 0; k < 3; k++) { bx.lo[k] = d_soa + (size_t)k * n; bx.hi[k] = d_soa + (size_t)(3 + k) * n; }
  KD_TRY(hipMemsetAsync(flag, 0, sizeof(uint32_t), st));

  // level 0: every primitive once, events sorted per axis
  hipLaunchKernelGGL(k_iota, grid(n), dim3(256), 0, st, (uint32_t)n, inst[0], task_of[0]);
  hipLaunchKernelGGL(k_ev_task0, grid(2 * n), dim3(256), 0, st, (uint32_t)(2 * n), ev_task[0]);
  for (int k = 0; k < 3; k++) {
    hipLaunchKernelGGL(k_make_events, grid(n), dim3(256), 0, st, bx, k, (uint32_t)n, keys_in, vals_in);
    size_t b2 = tmp_bytes;
    KD_TRY(rocprim::radix_sort_pairs(tmp, b2, keys_in, keys_out, vals_in, ev[0][k], 2 * n, 0, 64, st));
  }
  Task t0{};
  t0.start = 0; t0.count = (uint32_t)n;
  for (int k = 0; k < 3; k++) { t0.clo[k] = root.lo[k]; t0.chi[k] = root.hi[k]; }
  KD_TRY(hipMemcpyAsync(tasks[0], &t0, sizeof t0, hipMemcpyHostToDevice, st));

  struct Level { uint32_t first_node, nt; };
  std::vector<Level> levels;
  size_t nt = 1, ninst = n, nodes_total = 0, leaf_total = 0;
  int cur = 0;
  auto scan_u32 = [&](uint32_t* in, uint32_t* outp, size_t count) {
    size_t b2 = tmp_bytes;
    KD_TRY(rocprim::exclusive_scan(tmp, b2, in, outp, 0u, count, rocprim::plus<uint32_t>(), st));
  };
  auto scan_u64 = [&](unsigned long long* in, unsigned long long* outp, size_t count) {
    size_t b2 = tmp_bytes;
    KD_TRY(rocprim::exclusive_scan(tmp, b2, in, outp, 0ull, count, rocprim::plus<unsigned long long>(), st));
  };
  for (uint32_t depth = 0;; depth++) {
    // (no tree is too deep for the traversals since ABI v5 — the per-tree kernels' stacks are sized from the scene's
    // deepest tree and rpt_tree_generic walks the rest — so the build does not refuse one either.  The bound below only
    // ends a build that does not converge: the rule halves the primitives of a path every level or stops splitting, so
    // 64 levels would need 16 * 2^64 / 0.85^64 primitives.)
    if (depth > 64u) {
      why = "the device build did not converge (more than 64 levels)";
      return Attempt::failed;
    }
    if (nodes_total + nt > node_cap || leaf_total + ninst > leaf_cap) return Attempt::grow;
    const int nxt = cur ^ 1;
    hipLaunchKernelGGL(k_medians, grid(nt), dim3(256), 0, st, bx, tasks[cur], (uint32_t)nt, inst[cur], ev[cur][0], ev[cur][1],
                       ev[cur][2], dec);
    hipLaunchKernelGGL(k_scores, grid(ninst), dim3(256), 0, st, bx, tasks[cur], inst[cur], task_of[cur], (uint32_t)ninst, dec);
    hipLaunchKernelGGL(k_decide, grid(nt + 1), dim3(256), 0, st, tasks[cur], (uint32_t)nt, dec, is_split, leaf_count, child_count, flag);
    // (k_decide also zeroes the element past the end of each array: the exclusive scan's last entry is the total)
    scan_u32(is_split, split_rank, nt + 1);
    scan_u32(leaf_count, leaf_start, nt + 1);
    scan_u32(child_count, child_start, 2 * nt + 1);
    hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, st, split_rank + nt, leaf_start + nt, child_start + 2 * nt, d_totals);
    KD_TRY(hipMemcpyAsync(pinned, d_totals, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    KD_TRY(hipStreamSynchronize(st));
    const size_t n_split = pinned[0], n_leaf_inst = pinned[1], n_next = pinned[2];
    if (n_next > cap || 2 * n_split > task_cap) return Attempt::grow;
    hipLaunchKernelGGL(k_emit, grid(nt), dim3(256), 0, st, tasks[cur], (uint32_t)nt, dec, split_rank, child_start, leaf_start,
                       (uint32_t)leaf_total, d_nodes + nodes_total, tasks[nxt]);
    hipLaunchKernelGGL(k_flags, grid(ninst), dim3(256), 0, st, bx, dec, inst[cur], task_of[cur], (uint32_t)ninst, fl);
    scan_u64(fl, sfl, ninst);
    hipLaunchKernelGGL(k_scatter_inst, grid(ninst), dim3(256), 0, st, tasks[cur], dec, inst[cur], task_of[cur], (uint32_t)ninst,
                       fl, sfl, child_start, leaf_start, (uint32_t)leaf_total, inst[nxt], task_of[nxt], pos_l, pos_r, leaves);
    if (n_split) {
      for (int k = 0; k < 3; k++) {
        hipLaunchKernelGGL(k_ev_flags, grid(2 * ninst), dim3(256), 0, st, tasks[cur], task_of[cur], ev[cur][k],
                           (uint32_t)(2 * ninst), pos_l, pos_r, ev_task[cur], fl);
        scan_u64(fl, sfl, 2 * ninst);
        hipLaunchKernelGGL(k_ev_scatter, grid(2 * ninst), dim3(256), 0, st, tasks[cur], tasks[nxt], dec, ev[cur][k],
                           (uint32_t)(2 * ninst), pos_l, pos_r, ev_task[cur], fl, sfl, ev[nxt][k], ev_task[nxt]);
      }
    }
    levels.push_back({(uint32_t)nodes_total, (uint32_t)nt});
    nodes_total += nt;
    leaf_total += n_leaf_inst;
    if (!n_split) break;
    nt = 2 * n_split;
    ninst = n_next;
    cur = nxt;
  }
  KD_TRY(hipGetLastError());
  // ---- to the host: breadth-first nodes, leaf buffer; renumber depth-first as construct() numbers them
  std::vector<BfsNode> bfs(nodes_total);
  std::vector<uint32_t> leaf_host(std::max<size_t>(leaf_total, 1));
  KD_TRY(hipMemcpyAsync(bfs.data(), d_nodes, nodes_total * sizeof(BfsNode), hipMemcpyDeviceToHost, st));
  if (leaf_total) KD_TRY(hipMemcpyAsync(leaf_host.data(), leaves, leaf_total * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  KD_TRY(hipMemcpyAsync(pinned, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  KD_TRY(hipStreamSynchronize(st));
  const uint32_t irregular = pinned[0];

  out.nodes.clear(); out.refs.clear();
  out.nodes.reserve(nodes_total); out.refs.reserve(leaf_total);
  out.max_depth = (uint32_t)levels.size() - 1u;
  out.regular = irregular == 0;
  out.nodes.push_back({});
  struct Item { uint32_t level, index, dfs; };
  std::vector<Item> stack{{0u, 0u, 0u}};
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const BfsNode& b = bfs[levels[it.level].first_node + it.index];
    if ((b.info & 3u) == 3u) {
      const uint32_t cnt = b.info >> 2;
      rptdev::KdNode& nd = out.nodes[it.dfs];
      nd.split = 0.0; nd.a = (uint32_t)out.refs.size(); nd.ib = 3u | (cnt << 2);
      out.refs.insert(out.refs.end(), leaf_host.begin() + b.a, leaf_host.begin() + b.a + cnt);
    } else {
      const uint32_t l = (uint32_t)out.nodes.size();
      out.nodes.push_back({});
      out.nodes.push_back({});
      rptdev::KdNode& nd = out.nodes[it.dfs];
      nd.split = b.split; nd.a = l; nd.ib = b.info;
      // construct() builds the whole left subtree before the right one: the right child waits on the stack
      stack.push_back({it.level + 1u, b.a + 1u, l + 1u});
      stack.push_back({it.level + 1u, b.a, l});
    }
  }
  return Attempt::done;
}

} // namespace

// returns false (with `why`) when the device build cannot be used — the caller builds on the host then
bool kd_build_device(const std::vector<Box>& boxes, KdBuild& out, int device, std::string& why) {
  const size_t n = boxes.size();
  if (n < 16 || n >= (1ull << 28)) { why = "primitive count outside the device builder's range"; return false; }
  for (const Box& b : boxes)
    for (int k = 0; k < 3; k++)
      if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k])) { why = "non-finite box"; return false; }
  hipStream_t st = nullptr;
  uint32_t* pinned = nullptr;
  bool ok = false;
  int prev_device = -1; // the caller's current device is put back: a library call must not move it
  (void)hipGetDevice(&prev_device);
  // hipGetLastError() reports the thread's LAST failed runtime call, whoever made it: rocPRIM checks it after its launches
  // and would hand an old failure (another call's bad device ordinal, say) back as its own.  Start from a clean slate.
  (void)hipGetLastError();
  try {
    KD_TRY(hipSetDevice(device));
    KD_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    KD_TRY(hipHostMalloc((void**)&pinned, 4 * sizeof(uint32_t), hipHostMallocDefault));
    std::vector<double> soa(6 * n); // boxes, structure of arrays
    Box root;
    for (int k = 0; k < 3; k++) { root.lo[k] = INFINITY; root.hi[k] = -INFINITY; }
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        soa[(size_t)k * n + i] = boxes[i].lo[k];
        soa[(size_t)(3 + k) * n + i] = boxes[i].hi[k];
        root.lo[k] = std::fmin(root.lo[k], boxes[i].lo[k]);
        root.hi[k] = std::fmax(root.hi[k], boxes[i].hi[k]);
      }
    // a level holds more instances than primitives (straddlers go to both children): room for 6 n, doubled when a
    // level needs more (the reference rule duplicates ~4.5x on meshes)
    size_t cap = std::max<size_t>(6 * n, 4096);
    for (int attempt = 0; attempt < 6; attempt++, cap *= 2) {
      Attempt r = build_once(soa, n, root, cap, st, pinned, out, why);
      if (r == Attempt::done) { ok = true; break; }
      if (r == Attempt::failed) break;
      why = "the tree duplicates its primitives more than 190 times";
    }
  } catch (const HipErr& e) {
    (void)hipGetLastError();
    why = std::string(e.what) + ": " + hipGetErrorString(e.e);
    ok = false;
  } catch (const std::exception& e) { // (bad_alloc of a host vector: the stream and the pinned buffer are released below)
    why = std::string("device kd build: ") + e.what();
    ok = false;
  } catch (...) {
    why = "device kd build: unexpected exception";
    ok = false;
  }
  if (pinned) (void)hipHostFree(pinned);
  if (st) (void)hipStreamDestroy(st);
  if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);
  return ok;
}

} // namespace rpthost
