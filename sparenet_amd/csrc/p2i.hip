// p2i.hip -- differentiable point-to-image splat ("p2i") for MI355X (gfx950).
//
// Reference: cuda/p2i_op/p2i_max.h:7-143 (max fwd/bwd functors), p2i_sum.h:7-131
// (sum fwd/bwd), pixel walk utility.h:82-100, launcher common.h:96-115.
//
// Semantics (oracle/p2i.c): pixels x in [clamp(floor(px-R)), clamp(ceil(px+R))],
// same for y, r = sqrtf(dx*dx+dy*dy) <= R; weight = (float)(cos(r*pi/R)*0.5+0.5)
// with the cosine in DOUBLE; max: out = max(background, max_p feat*w) with strict
// '<' (a point merely equal to the current value does not replace it) and equal
// point values resolved to the LOWEST point id.
//
// MI355X design
//   * max forward is LOCK-FREE: the reference serialises every pixel hit through
//     a CAS spin-lock + three atomics; here one 64-bit global atomicMax carries
//     (order-preserving value bits << 32 | ~point id).  Background enters as
//     (value, 0xFFFFFFFF) so it wins value ties, lower ids win among points --
//     exactly the sequential semantics, deterministically.
//   * a point's footprint is spread over LPP lanes (LPP = 2^k >= box width): the
//     lanes of one point touch consecutive pixels of a row, so the 64-bit atomics
//     of a wave fall into few cache lines, and the double-precision cosine work
//     of a point (up to ~314 pixels at R=10) is shared by 16-32 lanes instead of
//     one thread.
#include "common.hpp"

namespace {

__device__ __forceinline__ unsigned ord_f32(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

struct Box {
  int min_x, max_x, min_y, max_y;
};
__device__ __forceinline__ Box box_of(float py, float px, int h, int w, float radius) {
  Box b;
  b.min_x = clampi((int)floorf(px - radius), 0, w - 1);
  b.max_x = clampi((int)ceilf(px + radius), 0, w - 1);
  b.min_y = clampi((int)floorf(py - radius), 0, h - 1);
  b.max_y = clampi((int)ceilf(py + radius), 0, h - 1);
  return b;
}

__device__ __forceinline__ float cos_weight(float r, float radius) {
  return (float)(cos((double)r * M_PI / (double)radius) * 0.5 + 0.5);
}

__device__ __forceinline__ float radius_of(float dx, float dy) {
#pragma clang fp contract(off)
  return __builtin_sqrtf(dx * dx + dy * dy);
}

__global__ __launch_bounds__(256) void p2i_max_init_kernel(const float *__restrict__ background,
                                                           unsigned long long *__restrict__ img,
                                                           long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    img[e] = ((unsigned long long)ord_f32(background[e]) << 32) | 0xFFFFFFFFull;
}

template <int LPP>
__global__ __launch_bounds__(256) void p2i_max_splat_kernel(
    const float *__restrict__ points, const float *__restrict__ feat,
    const int *__restrict__ batch_inds, unsigned long long *__restrict__ img, int npoints,
    int channels, int batch, int h, int w, float radius) {
  const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPP;
  const int l = threadIdx.x % LPP;
  if (gid >= (long)npoints * channels) return;
  const int c = (int)(gid % channels);
  const int pid = (int)((gid / channels) % npoints);
  const int b = batch_inds[pid];
  if (b < 0 || b >= batch) return;
  const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
  const float f = feat[gid];
  const Box bx = box_of(py, px, h, w, radius);
  const unsigned low = 0xFFFFFFFEu - (unsigned)pid;
  unsigned long long *plane = img + ((size_t)b * channels + c) * h * w;
  for (int x = bx.min_x + l; x <= bx.max_x; x += LPP) {
    const float dx = x - px;
    for (int y = bx.min_y; y <= bx.max_y; ++y) {
      const float dy = y - py;
      const float r = radius_of(dx, dy);
      if (r <= radius) {
        const float v = f * cos_weight(r, radius);
        const unsigned long long key = ((unsigned long long)ord_f32(v) << 32) | low;
        atomicMax(plane + (size_t)y * w + x, key);
      }
    }
  }
}

__global__ __launch_bounds__(256) void p2i_max_finalize_kernel(
    const unsigned long long *__restrict__ img, float *__restrict__ out, int *__restrict__ ids,
    long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const unsigned long long k = img[e];
    const unsigned low = (unsigned)k;
    out[e] = unord_f32((unsigned)(k >> 32));
    ids[e] = low == 0xFFFFFFFFu ? -1 : (int)(0xFFFFFFFEu - low);
  }
}

__global__ __launch_bounds__(256) void p2i_max_bwd_kernel(
    const float *__restrict__ out_grad, const int *__restrict__ out_ids,
    const float *__restrict__ points, const float *__restrict__ feat,
    float *__restrict__ points_grad, float *__restrict__ feat_grad,
    float *__restrict__ background_grad, int channels, int h, int w, float radius, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % w), y = (int)((e / w) % h);
    const int c = (int)((e / ((long)w * h)) % channels);
    const float g = out_grad[e];
    const int pid = out_ids[e];
    if (pid < 0) {
      background_grad[e] = g;
      continue;
    }
    background_grad[e] = 0.f;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const float dx = x - px, dy = y - py;
    const float r = radius_of(dx, dy);
    const float wgt = cos_weight(r, radius);
    const float fv = feat[(size_t)pid * channels + c];
    unsafeAtomicAdd(&feat_grad[(size_t)pid * channels + c], g * wgt);
    const float wg = g * fv;
    const float rm = r > 1e-10f ? r : 1e-10f;
    const float k = (float)((double)wg * sin((double)r * M_PI / (double)radius) * 0.5 * M_PI /
                            (double)radius / (double)rm);
    unsafeAtomicAdd(&points_grad[pid * 2 + 0], k * dy);
    unsafeAtomicAdd(&points_grad[pid * 2 + 1], k * dx);
  }
}

template <int LPP>
__global__ __launch_bounds__(256) void p2i_sum_fwd_kernel(
    const float *__restrict__ points, const float *__restrict__ feat,
    const int *__restrict__ batch_inds, float *__restrict__ out, int npoints, int channels,
    int batch, int h, int w, float radius) {
  const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPP;
  const int l = threadIdx.x % LPP;
  if (gid >= (long)npoints * channels) return;
  const int c = (int)(gid % channels);
  const int pid = (int)((gid / channels) % npoints);
  const int b = batch_inds[pid];
  if (b < 0 || b >= batch) return;
  const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
  const float f = feat[gid];
  const Box bx = box_of(py, px, h, w, radius);
  float *plane = out + ((size_t)b * channels + c) * h * w;
  for (int x = bx.min_x + l; x <= bx.max_x; x += LPP) {
    const float dx = x - px;
    for (int y = bx.min_y; y <= bx.max_y; ++y) {
      const float dy = y - py;
      const float r = radius_of(dx, dy);
      if (r <= radius) unsafeAtomicAdd(plane + (size_t)y * w + x, cos_weight(r, radius) * f);
    }
  }
}

// one thread per (point, channel), footprint walked in the reference's order
__global__ __launch_bounds__(256) void p2i_sum_bwd_kernel(
    const float *__restrict__ out_grad, const float *__restrict__ points,
    const float *__restrict__ feat, const int *__restrict__ batch_inds,
    float *__restrict__ points_grad, float *__restrict__ feat_grad, int npoints, int channels,
    int batch, int h, int w, float radius) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)npoints * channels) return;
  const int c = (int)(gid % channels);
  const int pid = (int)((gid / channels) % npoints);
  const int b = batch_inds[pid];
  float gf = 0.f, gy = 0.f, gx = 0.f;
  if (b >= 0 && b < batch) {
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const float fv = feat[gid];
    const Box bx = box_of(py, px, h, w, radius);
    const float *plane = out_grad + ((size_t)b * channels + c) * h * w;
    for (int x = bx.min_x; x <= bx.max_x; ++x) {
      const float dx = x - px;
      for (int y = bx.min_y; y <= bx.max_y; ++y) {
        const float dy = y - py;
        const float r = radius_of(dx, dy);
        if (!(r <= radius)) continue;
        const float wgt = cos_weight(r, radius);
        const float g = plane[(size_t)y * w + x];
        gf += g * wgt;
        const float wg = g * fv;
        const float rm = r > 1e-10f ? r : 1e-10f;
        const double s =
            (double)wg * sin((double)r * M_PI / (double)radius) * 0.5 * M_PI / (double)radius;
        gy += (float)(s * (double)dy / (double)rm);
        gx += (float)(s * (double)dx / (double)rm);
      }
    }
  }
  feat_grad[gid] = gf;
  if (channels == 1) {
    points_grad[pid * 2 + 0] = gy;
    points_grad[pid * 2 + 1] = gx;
  } else {
    unsafeAtomicAdd(&points_grad[pid * 2 + 0], gy);
    unsafeAtomicAdd(&points_grad[pid * 2 + 1], gx);
  }
}

int lanes_per_point(float radius) {
  const int width = (int)(2.f * (radius > 0.f ? radius : 0.f)) + 3;
  int l = 4;
  while (l < width && l < 64) l *= 2;
  return l;
}

int check_common(const char *fn, int npoints, int channels, int batch, int h, int w,
                 float radius) {
  if (!(npoints >= 0 && channels >= 1 && batch >= 1 && h >= 1 && w >= 1))
    return sn::fail(SN_EINVAL, "%s: bad sizes npoints=%d channels=%d batch=%d h=%d w=%d", fn,
                    npoints, channels, batch, h, w);
  if (!(radius > 0.f)) return sn::fail(SN_EINVAL, "%s: kernel radius must be > 0", fn);
  if ((long)batch * channels * h * w >= (1L << 31) || (long)npoints * channels >= (1L << 31))
    return sn::fail(SN_EINVAL, "%s: tensor too large", fn);
  return 0;
}

int lin_blocks(long total) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t sn_p2i_max_workspace_bytes(int batch, int channels, int h, int w) {
  if (batch < 1 || channels < 1 || h < 1 || w < 1) return 0;
  return (size_t)batch * channels * h * w * 8;
}

extern "C" int sn_p2i_max_forward(const float *points, const float *feat, const int *batch_inds,
                                  const float *background, int npoints, int channels, int batch,
                                  int h, int w, float radius, float *out, int *out_ids,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  SN_REQUIRE(background && out && out_ids && workspace, "sn_p2i_max_forward: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds), "sn_p2i_max_forward: null pointer");
  if (int rc = check_common("sn_p2i_max_forward", npoints, channels, batch, h, w, radius)) return rc;
  SN_REQUIRE(workspace_bytes >= sn_p2i_max_workspace_bytes(batch, channels, h, w),
             "sn_p2i_max_forward: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  unsigned long long *img = static_cast<unsigned long long *>(workspace);
  const long px = (long)batch * channels * h * w;
  p2i_max_init_kernel<<<lin_blocks(px), 256, 0, s>>>(background, img, px);
  const long groups = (long)npoints * channels;
  if (groups > 0) {
    const int lpp = lanes_per_point(radius);
    const long blocks = (groups * lpp + 255) / 256;
    SN_REQUIRE(blocks < (1L << 31), "sn_p2i_max_forward: too many points");
#define SN_SPLAT(L)                                                                          \
  p2i_max_splat_kernel<L><<<(int)blocks, 256, 0, s>>>(points, feat, batch_inds, img, npoints, \
                                                      channels, batch, h, w, radius)
    if (sn::prof_enabled()) sn::prof_begin("p2i_max_splat", s);
    switch (lpp) {
      case 4: SN_SPLAT(4); break;
      case 8: SN_SPLAT(8); break;
      case 16: SN_SPLAT(16); break;
      case 32: SN_SPLAT(32); break;
      default: SN_SPLAT(64); break;
    }
    if (sn::prof_enabled()) sn::prof_end("p2i_max_splat", s);
#undef SN_SPLAT
  }
  p2i_max_finalize_kernel<<<lin_blocks(px), 256, 0, s>>>(img, out, out_ids, px);
  return sn::launch_status("sn_p2i_max_forward");
}

extern "C" int sn_p2i_max_backward(const float *out_grad, const int *out_ids, const float *points,
                                   const float *feat, int npoints, int channels, int batch, int h,
                                   int w, float radius, float *points_grad, float *feat_grad,
                                   float *background_grad, void *stream) {
  SN_REQUIRE(out_grad && out_ids && background_grad, "sn_p2i_max_backward: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && points_grad && feat_grad),
             "sn_p2i_max_backward: null pointer");
  if (int rc = check_common("sn_p2i_max_backward", npoints, channels, batch, h, w, radius)) return rc;
  hipStream_t s = sn::as_stream(stream);
  if (npoints > 0) {
    SN_HIP(hipMemsetAsync(points_grad, 0, (size_t)npoints * 2 * 4, s));
    SN_HIP(hipMemsetAsync(feat_grad, 0, (size_t)npoints * channels * 4, s));
  }
  const long px = (long)batch * channels * h * w;
  p2i_max_bwd_kernel<<<lin_blocks(px), 256, 0, s>>>(out_grad, out_ids, points, feat, points_grad,
                                                    feat_grad, background_grad, channels, h, w,
                                                    radius, px);
  return sn::launch_status("sn_p2i_max_backward");
}

extern "C" int sn_p2i_sum_forward(const float *points, const float *feat, const int *batch_inds,
                                  int npoints, int channels, int batch, int h, int w, float radius,
                                  float *out, void *stream) {
  SN_REQUIRE(out, "sn_p2i_sum_forward: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds), "sn_p2i_sum_forward: null pointer");
  if (int rc = check_common("sn_p2i_sum_forward", npoints, channels, batch, h, w, radius)) return rc;
  const long groups = (long)npoints * channels;
  if (groups == 0) return 0;
  hipStream_t s = sn::as_stream(stream);
  const int lpp = lanes_per_point(radius);
  const long blocks = (groups * lpp + 255) / 256;
  SN_REQUIRE(blocks < (1L << 31), "sn_p2i_sum_forward: too many points");
#define SN_SUM(L)                                                                            \
  p2i_sum_fwd_kernel<L><<<(int)blocks, 256, 0, s>>>(points, feat, batch_inds, out, npoints,   \
                                                    channels, batch, h, w, radius)
  switch (lpp) {
    case 4: SN_SUM(4); break;
    case 8: SN_SUM(8); break;
    case 16: SN_SUM(16); break;
    case 32: SN_SUM(32); break;
    default: SN_SUM(64); break;
  }
#undef SN_SUM
  return sn::launch_status("sn_p2i_sum_forward");
}

extern "C" int sn_p2i_sum_backward(const float *out_grad, const float *points, const float *feat,
                                   const int *batch_inds, int npoints, int channels, int batch,
                                   int h, int w, float radius, float *points_grad,
                                   float *feat_grad, void *stream) {
  SN_REQUIRE(out_grad, "sn_p2i_sum_backward: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds && points_grad && feat_grad),
             "sn_p2i_sum_backward: null pointer");
  if (int rc = check_common("sn_p2i_sum_backward", npoints, channels, batch, h, w, radius)) return rc;
  const long groups = (long)npoints * channels;
  if (groups == 0) return 0;
  hipStream_t s = sn::as_stream(stream);
  if (channels > 1) SN_HIP(hipMemsetAsync(points_grad, 0, (size_t)npoints * 2 * 4, s));
  p2i_sum_bwd_kernel<<<(int)((groups + 255) / 256), 256, 0, s>>>(
      out_grad, points, feat, batch_inds, points_grad, feat_grad, npoints, channels, batch, h, w,
      radius);
  return sn::launch_status("sn_p2i_sum_backward");
}
