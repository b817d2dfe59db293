// api_buffer.cpp — the device-resident Buffer (buffer.rs:9-93 on the device: add_samples, image, variance; see api_internal.h)
#include "api_internal.h"

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
