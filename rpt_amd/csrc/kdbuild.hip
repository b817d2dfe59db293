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

template <class T> struct Dev {
  T* p = nullptr;
  size_t n = 0;
  Dev() = default;
  Dev(const Dev&) = delete;
  Dev& operator=(const Dev&) = delete;
  ~Dev() { if (p) (void)hipFree(p); }
  void need(size_t count) { // grow-only, contents NOT kept; half as much again so that growing levels do not realloc each time
    if (count <= n && p) return;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    count = std::max<size_t>(count + count / 2, 16);
    KD_TRY(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
};

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
  keys[2 * i] = b.lo[k][i]; vals[2 * i] = i << 1;
  keys[2 * i + 1] = b.hi[k][i]; vals[2 * i + 1] = (i << 1) | 1u;
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

inline dim3 grid(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

} // namespace

// returns false (with `why`) when the device build cannot be used — the caller builds on the host then
bool kd_build_device(const std::vector<Box>& boxes, KdBuild& out, int device, std::string& why) {
  const size_t n = boxes.size();
  if (n < 16 || n >= (1ull << 30)) { why = "primitive count outside the device builder's range"; return false; }
  for (const Box& b : boxes)
    for (int k = 0; k < 3; k++)
      if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k])) { why = "non-finite box"; return false; }
  try {
    KD_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    KD_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { if (s) (void)hipStreamDestroy(s); } } guard{st};

    // boxes, structure of arrays
    std::vector<double> soa(6 * n);
    Box root;
    for (int k = 0; k < 3; k++) { root.lo[k] = INFINITY; root.hi[k] = -INFINITY; }
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        soa[(size_t)k * n + i] = boxes[i].lo[k];
        soa[(size_t)(3 + k) * n + i] = boxes[i].hi[k];
        root.lo[k] = std::fmin(root.lo[k], boxes[i].lo[k]);
        root.hi[k] = std::fmax(root.hi[k], boxes[i].hi[k]);
      }
    Dev<double> d_soa;
    d_soa.need(6 * n);
    KD_TRY(hipMemcpyAsync(d_soa.p, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, st));
    Boxes bx;
    for (int k = 0; k < 3; k++) { bx.lo[k] = d_soa.p + (size_t)k * n; bx.hi[k] = d_soa.p + (size_t)(3 + k) * n; }

    // level buffers, double-buffered, sized per level (a level holds more instances than the one above: straddlers)
    Dev<uint32_t> inst[2], task_of[2], ev[2][3], ev_task[2][3], pos_l, pos_r, leaves;
    Dev<unsigned long long> fl, sfl;
    Dev<Task> tasks[2];
    Dev<Decision> dec;
    Dev<uint32_t> is_split, split_rank, leaf_count, leaf_start, child_count, child_start, flag;
    Dev<BfsNode> d_nodes;
    Dev<uint8_t> tmp;
    auto size_level = [&](int which, size_t c) {
      inst[which].need(c); task_of[which].need(c);
      for (int k = 0; k < 3; k++) { ev[which][k].need(2 * c); ev_task[which][k].need(2 * c); }
    };
    size_level(0, n);
    flag.need(1);
    KD_TRY(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), st));

    // level 0: every primitive once, events sorted per axis
    {
      Dev<double> keys_in, keys_out;
      Dev<uint32_t> vals_in;
      keys_in.need(2 * n); keys_out.need(2 * n); vals_in.need(2 * n);
      size_t bytes = 0;
      KD_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.p, keys_out.p, vals_in.p, ev[0][0].p, 2 * n, 0, 64, st));
      tmp.need(bytes);
      hipLaunchKernelGGL(k_iota, grid(n), dim3(256), 0, st, (uint32_t)n, inst[0].p, task_of[0].p);
      for (int k = 0; k < 3; k++) {
        hipLaunchKernelGGL(k_make_events, grid(n), dim3(256), 0, st, bx, k, (uint32_t)n, keys_in.p, vals_in.p);
        size_t b2 = tmp.n;
        KD_TRY(rocprim::radix_sort_pairs(tmp.p, b2, keys_in.p, keys_out.p, vals_in.p, ev[0][k].p, 2 * n, 0, 64, st));
        hipLaunchKernelGGL(k_ev_task0, grid(2 * n), dim3(256), 0, st, (uint32_t)(2 * n), ev_task[0][k].p);
      }
      KD_TRY(hipStreamSynchronize(st)); // keys_* die with this block
    }
    Task t0{};
    t0.start = 0; t0.count = (uint32_t)n;
    for (int k = 0; k < 3; k++) { t0.clo[k] = root.lo[k]; t0.chi[k] = root.hi[k]; }
    tasks[0].need(1);
    KD_TRY(hipMemcpyAsync(tasks[0].p, &t0, sizeof t0, hipMemcpyHostToDevice, st));

    struct Level { uint32_t first_node, nt; };
    std::vector<Level> levels;
    size_t nt = 1, ninst = n, nodes_total = 0, leaf_total = 0;
    int cur = 0;
    auto scan_u32 = [&](uint32_t* in, uint32_t* outp, size_t count) {
      size_t bytes = 0;
      KD_TRY(rocprim::exclusive_scan(nullptr, bytes, in, outp, 0u, count, rocprim::plus<uint32_t>(), st));
      tmp.need(bytes);
      bytes = tmp.n;
      KD_TRY(rocprim::exclusive_scan(tmp.p, bytes, in, outp, 0u, count, rocprim::plus<uint32_t>(), st));
    };
    auto scan_u64 = [&](unsigned long long* in, unsigned long long* outp, size_t count) {
      size_t bytes = 0;
      KD_TRY(rocprim::exclusive_scan(nullptr, bytes, in, outp, 0ull, count, rocprim::plus<unsigned long long>(), st));
      tmp.need(bytes);
      bytes = tmp.n;
      KD_TRY(rocprim::exclusive_scan(tmp.p, bytes, in, outp, 0ull, count, rocprim::plus<unsigned long long>(), st));
    };
    uint32_t depth = 0;
    for (;; depth++) {
      if (depth > (uint32_t)rptdev::KD_MAX_STACK + 1) { // the host builder will say so properly (RPTGPU_E_TREE_TOO_DEEP)
        why = "tree deeper than the device traversal stack";
        return false;
      }
      const int nxt = cur ^ 1;
      dec.need(nt); is_split.need(nt + 1); split_rank.need(nt + 1); leaf_count.need(nt + 1); leaf_start.need(nt + 1);
      child_count.need(2 * nt + 1); child_start.need(2 * nt + 1);
      if (nodes_total + nt > d_nodes.n) { // grow, keeping what is there
        Dev<BfsNode> bigger;
        bigger.need(2 * (nodes_total + nt));
        if (nodes_total) KD_TRY(hipMemcpyAsync(bigger.p, d_nodes.p, nodes_total * sizeof(BfsNode), hipMemcpyDeviceToDevice, st));
        KD_TRY(hipStreamSynchronize(st));
        std::swap(bigger.p, d_nodes.p); std::swap(bigger.n, d_nodes.n);
      }
      if (leaf_total + ninst > leaves.n) {
        Dev<uint32_t> bigger;
        bigger.need(2 * (leaf_total + ninst));
        if (leaf_total) KD_TRY(hipMemcpyAsync(bigger.p, leaves.p, leaf_total * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        KD_TRY(hipStreamSynchronize(st));
        std::swap(bigger.p, leaves.p); std::swap(bigger.n, leaves.n);
      }
      hipLaunchKernelGGL(k_medians, grid(nt), dim3(256), 0, st, bx, tasks[cur].p, (uint32_t)nt, inst[cur].p, ev[cur][0].p,
                         ev[cur][1].p, ev[cur][2].p, dec.p);
      hipLaunchKernelGGL(k_scores, grid(ninst), dim3(256), 0, st, bx, tasks[cur].p, inst[cur].p, task_of[cur].p,
                         (uint32_t)ninst, dec.p);
      hipLaunchKernelGGL(k_decide, grid(nt), dim3(256), 0, st, tasks[cur].p, (uint32_t)nt, dec.p, is_split.p, leaf_count.p,
                         child_count.p, flag.p);
      // one extra element: the exclusive scan's last entry is the total
      KD_TRY(hipMemsetAsync(is_split.p + nt, 0, sizeof(uint32_t), st));
      KD_TRY(hipMemsetAsync(leaf_count.p + nt, 0, sizeof(uint32_t), st));
      KD_TRY(hipMemsetAsync(child_count.p + 2 * nt, 0, sizeof(uint32_t), st));
      scan_u32(is_split.p, split_rank.p, nt + 1);
      scan_u32(leaf_count.p, leaf_start.p, nt + 1);
      scan_u32(child_count.p, child_start.p, 2 * nt + 1);
      uint32_t totals[3];
      KD_TRY(hipMemcpyAsync(&totals[0], split_rank.p + nt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      KD_TRY(hipMemcpyAsync(&totals[1], leaf_start.p + nt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      KD_TRY(hipMemcpyAsync(&totals[2], child_start.p + 2 * nt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      KD_TRY(hipStreamSynchronize(st));
      const size_t n_split = totals[0], n_leaf_inst = totals[1], n_next = totals[2];
      size_level(nxt, std::max<size_t>(n_next, 1)); // (the other half of the double buffer: about to be overwritten)
      pos_l.need(ninst); pos_r.need(ninst); fl.need(2 * ninst); sfl.need(2 * ninst);
      tasks[nxt].need(std::max<size_t>(2 * n_split, 1));
      hipLaunchKernelGGL(k_emit, grid(nt), dim3(256), 0, st, tasks[cur].p, (uint32_t)nt, dec.p, split_rank.p, child_start.p,
                         leaf_start.p, (uint32_t)leaf_total, d_nodes.p + nodes_total, tasks[nxt].p);
      hipLaunchKernelGGL(k_flags, grid(ninst), dim3(256), 0, st, bx, dec.p, inst[cur].p, task_of[cur].p, (uint32_t)ninst, fl.p);
      scan_u64(fl.p, sfl.p, ninst);
      hipLaunchKernelGGL(k_scatter_inst, grid(ninst), dim3(256), 0, st, tasks[cur].p, dec.p, inst[cur].p, task_of[cur].p,
                         (uint32_t)ninst, fl.p, sfl.p, child_start.p, leaf_start.p, (uint32_t)leaf_total, inst[nxt].p,
                         task_of[nxt].p, pos_l.p, pos_r.p, leaves.p);
      if (n_split) {
        for (int k = 0; k < 3; k++) {
          hipLaunchKernelGGL(k_ev_flags, grid(2 * ninst), dim3(256), 0, st, tasks[cur].p, task_of[cur].p, ev[cur][k].p,
                             (uint32_t)(2 * ninst), pos_l.p, pos_r.p, ev_task[cur][k].p, fl.p);
          scan_u64(fl.p, sfl.p, 2 * ninst);
          hipLaunchKernelGGL(k_ev_scatter, grid(2 * ninst), dim3(256), 0, st, tasks[cur].p, tasks[nxt].p, dec.p, ev[cur][k].p,
                             (uint32_t)(2 * ninst), pos_l.p, pos_r.p, ev_task[cur][k].p, fl.p, sfl.p, ev[nxt][k].p,
                             ev_task[nxt][k].p);
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
    uint32_t irregular = 0;
    KD_TRY(hipMemcpyAsync(bfs.data(), d_nodes.p, nodes_total * sizeof(BfsNode), hipMemcpyDeviceToHost, st));
    if (leaf_total) KD_TRY(hipMemcpyAsync(leaf_host.data(), leaves.p, leaf_total * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    KD_TRY(hipMemcpyAsync(&irregular, flag.p, sizeof irregular, hipMemcpyDeviceToHost, st));
    KD_TRY(hipStreamSynchronize(st));

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
      rptdev::KdNode& nd = out.nodes[it.dfs];
      if ((b.info & 3u) == 3u) {
        const uint32_t cnt = b.info >> 2;
        nd.split = 0.0; nd.a = (uint32_t)out.refs.size(); nd.ib = 3u | (cnt << 2);
        out.refs.insert(out.refs.end(), leaf_host.begin() + b.a, leaf_host.begin() + b.a + cnt);
      } else {
        const uint32_t l = (uint32_t)out.nodes.size();
        nd.split = b.split; nd.a = l; nd.ib = b.info;
        out.nodes.push_back({});
        out.nodes.push_back({});
        // construct() builds the whole left subtree before the right one: the right child waits on the stack
        stack.push_back({it.level + 1u, b.a + 1u, l + 1u});
        stack.push_back({it.level + 1u, b.a, l});
      }
    }
    return true;
  } catch (const HipErr& e) {
    (void)hipGetLastError();
    why = std::string(e.what) + ": " + hipGetErrorString(e.e);
    return false;
  }
}

} // namespace rpthost
