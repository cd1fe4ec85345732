// p2i_f64.hip -- the point-to-image splat for float64 tensors (gfx950).
//
// The reference dispatches its p2i functors on float and double (cuda/p2i_op/p2i_max.h:177,218,
// p2i_sum.h:162,201) and its only test is a float64 gradcheck (cuda/p2i_op/p2i_test.py:23-35).  SpareNet
// itself renders in fp32 (p2i.hip holds that path: binned gather, fixed-point backward); this file is the
// double-precision surface, written for exactness and simplicity, not speed:
//   max forward   two passes over (point, channel): 64-bit atomicMax on an order-preserving key of the
//                 value, then atomicMin of the point id among the points attaining it -- the reference's
//                 "strictly greater replaces" rule with the lowest id on ties (its GPU order is a race)
//   max backward  one thread per pixel, double atomicAdd to the winner (p2i_max.h:68-143)
//   sum forward   double atomicAdd per footprint pixel (p2i_sum.h:7-58)
//   sum backward  one thread per (point, channel) walks its own footprint: no atomics except on the
//                 point's two coordinates shared by its channels (p2i_sum.h:60-131)
// Pixel walk and weight exactly as cuda/p2i_op/utility.h:82-100: x outer / y inner over
// clamp(floor(p - R)) .. clamp(ceil(p + R)), r = sqrt(dx^2 + dy^2) <= R, w = cos(r pi / R) / 2 + 1/2.
#include "common.hpp"

namespace {

constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ int clampi64(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ unsigned long long ord_f64(double v) {  // monotone double -> u64
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unord_f64(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

template <typename F>
__device__ __forceinline__ void for_each_pixel(double py, double px, int h, int w, double radius, F f) {
  if (!(fabs(py) < 1e15) || !(fabs(px) < 1e15)) return;  // NaN / inf: int conversion undefined
  const int min_x = clampi64((int)floor(px - radius), 0, w - 1), max_x = clampi64((int)ceil(px + radius), 0, w - 1);
  const int min_y = clampi64((int)floor(py - radius), 0, h - 1), max_y = clampi64((int)ceil(py + radius), 0, h - 1);
  for (int x = min_x; x <= max_x; ++x)
    for (int y = min_y; y <= max_y; ++y) {
      const double dx = x - px, dy = y - py;
      const double r = sqrt(dx * dx + dy * dy);
      if (r <= radius) f(y, x, dy, dx, r);
    }
}

__device__ __forceinline__ double cos_weight64(double r, double radius) { return cos(r * kPi / radius) * 0.5 + 0.5; }

__global__ __launch_bounds__(256) void f64_init_kernel(const double *__restrict__ bg, long total,
                                                      unsigned long long *__restrict__ key, int *__restrict__ ids) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    key[e] = ord_f64(bg[e]);
    ids[e] = 0x7fffffff;
  }
}

template <int PASS>
__global__ __launch_bounds__(256) void f64_max_fwd_kernel(const double *__restrict__ points,
                                                         const double *__restrict__ feat,
                                                         const int *__restrict__ batch_inds,
                                                         const double *__restrict__ bg, int npoints, int channels,
                                                         int batch, int h, int w, double radius,
                                                         unsigned long long *__restrict__ key, int *__restrict__ ids) {
  const long total = (long)npoints * channels;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % channels), pid = (int)(id / channels);
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const double f = feat[id];
    for_each_pixel(points[pid * 2], points[pid * 2 + 1], h, w, radius, [&](int y, int x, double, double, double r) {
      const size_t index = (((size_t)b * channels + c) * h + y) * w + x;
      const double v = f * cos_weight64(r, radius);
      const unsigned long long k = ord_f64(v);
      if (PASS == 0) {
        atomicMax(&key[index], k);
      } else if (k == key[index] && bg[index] < v) {  // attains the maximum, and the maximum beat the background
        atomicMin(&ids[index], pid);
      }
    });
  }
}

__global__ __launch_bounds__(256) void f64_max_finish_kernel(const unsigned long long *__restrict__ key, long total,
                                                            double *__restrict__ out, int *__restrict__ ids) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    out[e] = unord_f64(key[e]);
    if (ids[e] == 0x7fffffff) ids[e] = -1;
  }
}

__global__ __launch_bounds__(256) void f64_max_bwd_kernel(const double *__restrict__ out_grad,
                                                         const int *__restrict__ ids,
                                                         const double *__restrict__ points,
                                                         const double *__restrict__ feat, int channels, int batch,
                                                         int h, int w, double radius, double *__restrict__ gp,
                                                         double *__restrict__ gf, double *__restrict__ gb) {
  const long total = (long)batch * channels * h * w;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % w), y = (int)((e / w) % h), c = (int)((e / ((long)w * h)) % channels);
    const double g = out_grad[e];
    const int pid = ids[e];
    if (pid < 0) {
      gb[e] = g;
      continue;
    }
    gb[e] = 0.0;
    const double px = points[pid * 2 + 1], py = points[pid * 2];
    const double dx = x - px, dy = y - py;
    const double r = sqrt(dx * dx + dy * dy);
    const double wgt = cos_weight64(r, radius);
    const double fv = feat[(size_t)pid * channels + c];
    atomicAdd(&gf[(size_t)pid * channels + c], g * wgt);
    const double k = g * fv * sin(r * kPi / radius) * 0.5 * kPi / radius / fmax(r, 1e-10);
    atomicAdd(&gp[pid * 2], k * dy);
    atomicAdd(&gp[pid * 2 + 1], k * dx);
  }
}

__global__ __launch_bounds__(256) void f64_sum_fwd_kernel(const double *__restrict__ points,
                                                         const double *__restrict__ feat,
                                                         const int *__restrict__ batch_inds, int npoints,
                                                         int channels, int batch, int h, int w, double radius,
                                                         double *__restrict__ out) {
  const long total = (long)npoints * channels;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % channels), pid = (int)(id / channels);
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const double f = feat[id];
    for_each_pixel(points[pid * 2], points[pid * 2 + 1], h, w, radius, [&](int y, int x, double, double, double r) {
      atomicAdd(&out[(((size_t)b * channels + c) * h + y) * w + x], cos_weight64(r, radius) * f);
    });
  }
}

__global__ __launch_bounds__(256) void f64_sum_bwd_kernel(const double *__restrict__ out_grad,
                                                         const double *__restrict__ points,
                                                         const double *__restrict__ feat,
                                                         const int *__restrict__ batch_inds, int npoints,
                                                         int channels, int batch, int h, int w, double radius,
                                                         double *__restrict__ gp, double *__restrict__ gf) {
  const long total = (long)npoints * channels;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int c = (int)(id % channels), pid = (int)(id / channels);
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const double f = feat[id];
    double af = 0.0, ay = 0.0, ax = 0.0;
    for_each_pixel(points[pid * 2], points[pid * 2 + 1], h, w, radius, [&](int y, int x, double dy, double dx, double r) {
      const double g = out_grad[(((size_t)b * channels + c) * h + y) * w + x];
      af += g * cos_weight64(r, radius);
      const double k = g * f * sin(r * kPi / radius) * 0.5 * kPi / radius / fmax(r, 1e-10);
      ay += k * dy;
      ax += k * dx;
    });
    gf[id] = af;
    atomicAdd(&gp[pid * 2], ay);       // the channels of a point share its coordinates
    atomicAdd(&gp[pid * 2 + 1], ax);
  }
}

int blocks_of(long n) {
  const long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int check64(const char *fn, int npoints, int channels, int batch, int h, int w, double radius) {
  SN_REQUIRE(npoints >= 0 && channels >= 1 && batch >= 1 && h >= 1 && w >= 1, "%s: bad sizes", fn);
  SN_REQUIRE(radius > 0.0, "%s: kernel_radius must be positive", fn);
  SN_REQUIRE((long)batch * channels * h * w < (1L << 31), "%s: image tensor too large", fn);
  return 0;
}

}  // namespace

extern "C" size_t sn_p2i_f64_workspace_bytes(int batch, int channels, int h, int w) {
  if (batch < 1 || channels < 1 || h < 1 || w < 1) return 0;
  return (size_t)batch * channels * h * w * 8;
}

extern "C" int sn_p2i_max_forward_f64(const double *points, const double *feat, const int *batch_inds,
                                      const double *background, int npoints, int channels, int batch, int h,
                                      int w, double radius, double *out, int *out_ids, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  SN_REQUIRE(background && out && out_ids && workspace, "sn_p2i_max_forward_f64: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds), "sn_p2i_max_forward_f64: null pointer");
  if (int rc = check64("sn_p2i_max_forward_f64", npoints, channels, batch, h, w, radius)) return rc;
  SN_REQUIRE(workspace_bytes >= sn_p2i_f64_workspace_bytes(batch, channels, h, w),
             "sn_p2i_max_forward_f64: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  const long total = (long)batch * channels * h * w;
  unsigned long long *key = static_cast<unsigned long long *>(workspace);
  f64_init_kernel<<<blocks_of(total), 256, 0, s>>>(background, total, key, out_ids);
  const long groups = (long)npoints * channels;
  if (groups > 0) {
    f64_max_fwd_kernel<0><<<blocks_of(groups), 256, 0, s>>>(points, feat, batch_inds, background, npoints,
                                                            channels, batch, h, w, radius, key, out_ids);
    f64_max_fwd_kernel<1><<<blocks_of(groups), 256, 0, s>>>(points, feat, batch_inds, background, npoints,
                                                            channels, batch, h, w, radius, key, out_ids);
  }
  f64_max_finish_kernel<<<blocks_of(total), 256, 0, s>>>(key, total, out, out_ids);
  return sn::launch_status("sn_p2i_max_forward_f64");
}

extern "C" int sn_p2i_max_backward_f64(const double *out_grad, const int *out_ids, const double *points,
                                       const double *feat, int npoints, int channels, int batch, int h, int w,
                                       double radius, double *points_grad, double *feat_grad,
                                       double *background_grad, void *stream) {
  SN_REQUIRE(out_grad && out_ids && background_grad, "sn_p2i_max_backward_f64: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && points_grad && feat_grad), "sn_p2i_max_backward_f64: null pointer");
  if (int rc = check64("sn_p2i_max_backward_f64", npoints, channels, batch, h, w, radius)) return rc;
  hipStream_t s = sn::as_stream(stream);
  if (npoints > 0) {
    SN_HIP(hipMemsetAsync(points_grad, 0, (size_t)npoints * 2 * 8, s));
    SN_HIP(hipMemsetAsync(feat_grad, 0, (size_t)npoints * channels * 8, s));
  }
  const long total = (long)batch * channels * h * w;
  f64_max_bwd_kernel<<<blocks_of(total), 256, 0, s>>>(out_grad, out_ids, points, feat, channels, batch, h, w,
                                                      radius, points_grad, feat_grad, background_grad);
  return sn::launch_status("sn_p2i_max_backward_f64");
}

extern "C" int sn_p2i_sum_forward_f64(const double *points, const double *feat, const int *batch_inds, int npoints,
                                      int channels, int batch, int h, int w, double radius, double *out,
                                      void *stream) {
  SN_REQUIRE(out, "sn_p2i_sum_forward_f64: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds), "sn_p2i_sum_forward_f64: null pointer");
  if (int rc = check64("sn_p2i_sum_forward_f64", npoints, channels, batch, h, w, radius)) return rc;
  const long groups = (long)npoints * channels;
  if (groups == 0) return 0;
  f64_sum_fwd_kernel<<<blocks_of(groups), 256, 0, sn::as_stream(stream)>>>(points, feat, batch_inds, npoints,
                                                                          channels, batch, h, w, radius, out);
  return sn::launch_status("sn_p2i_sum_forward_f64");
}

extern "C" int sn_p2i_sum_backward_f64(const double *out_grad, const double *points, const double *feat,
                                       const int *batch_inds, int npoints, int channels, int batch, int h, int w,
                                       double radius, double *points_grad, double *feat_grad, void *stream) {
  SN_REQUIRE(out_grad, "sn_p2i_sum_backward_f64: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds && points_grad && feat_grad),
             "sn_p2i_sum_backward_f64: null pointer");
  if (int rc = check64("sn_p2i_sum_backward_f64", npoints, channels, batch, h, w, radius)) return rc;
  const long groups = (long)npoints * channels;
  if (groups == 0) return 0;
  hipStream_t s = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(points_grad, 0, (size_t)npoints * 2 * 8, s));
  SN_HIP(hipMemsetAsync(feat_grad, 0, (size_t)npoints * channels * 8, s));
  f64_sum_bwd_kernel<<<blocks_of(groups), 256, 0, s>>>(out_grad, points, feat, batch_inds, npoints, channels,
                                                       batch, h, w, radius, points_grad, feat_grad);
  return sn::launch_status("sn_p2i_sum_backward_f64");
}
