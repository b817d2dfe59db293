// api_common.cpp — errors, the RCCL loader, kernel tables; version, strerror, device count, stats (see api_internal.h)
#include "api_internal.h"

namespace rptapi {

thread_local std::string g_create_error;

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

int fail(rptgpu_scene* h, int code, const std::string& detail) {
  if (h) h->error = detail;
  else g_create_error = detail;
  return code;
}

int hip_fail(rptgpu_scene* h, const HipError& e) {
  char buf[512];
  std::snprintf(buf, sizeof buf, "%s failed at %s:%d: %s", e.what, e.file, e.line, hipGetErrorString(e.e));
  int code = (e.e == hipErrorOutOfMemory) ? RPTGPU_E_OUT_OF_MEMORY
             : (e.e == hipErrorNoDevice || e.e == hipErrorInvalidDevice || e.e == hipErrorInsufficientDriver)
                 ? RPTGPU_E_NO_DEVICE
                 : RPTGPU_E_HIP;
  return fail(h, code, buf);
}

const KernelTable* table_for(uint32_t /*mode: RPT_PRECISION_F64_STRICT is the only one*/, bool ext) {
  return ext ? &rpt_strict_ext::TABLE : &rpt_strict::TABLE;
}
const char* const BAD_MODE = "unknown precision_mode (RPT_PRECISION_F64_STRICT = 0 is the only mode; F64_FAST was removed in ABI v4)";

} // namespace rptapi

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
