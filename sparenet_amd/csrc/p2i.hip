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
#include <cmath>
#include <cstring>

#include "common.hpp"
#include "wave_dpp.hpp"

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

// The exact weight (float)(cos(pi r / R) / 2 + 1 / 2) with the cosine in double, as the reference evaluates it -- through
// cos(pi t) = -sin(pi (t - 1/2)), t = r / R in [0, 1], and the sine's Taylor series on [-pi/2, pi/2] (12 terms,
// within 2.3e-16 of the double cosine over the whole range: the library's cos() with its general range reduction
// and the fp64 division in front of it were most of the gather's epilogue).  inv_radius = 1.0 / (double)R.
__device__ __forceinline__ float cos_weight_inv(float r, double inv_radius) {
  const double y = ((double)r * inv_radius - 0.5) * M_PI;
  const double y2 = y * y;
  double p = -3.868170170630684e-23;
  p = __builtin_fma(p, y2, 1.9572941063391263e-20);
  p = __builtin_fma(p, y2, -8.22063524662433e-18);
  p = __builtin_fma(p, y2, 2.8114572543455206e-15);
  p = __builtin_fma(p, y2, -7.647163731819816e-13);
  p = __builtin_fma(p, y2, 1.6059043836821613e-10);
  p = __builtin_fma(p, y2, -2.505210838544172e-08);
  p = __builtin_fma(p, y2, 2.7557319223985893e-06);
  p = __builtin_fma(p, y2, -0.0001984126984126984);
  p = __builtin_fma(p, y2, 0.008333333333333333);
  p = __builtin_fma(p, y2, -0.16666666666666666);
  p = __builtin_fma(p, y2, 1.0);
  return (float)(0.5 - 0.5 * (y * p));
}
__device__ __forceinline__ float cos_weight(float r, float radius) {
  return cos_weight_inv(r, 1.0 / (double)radius);
}

__device__ __forceinline__ float sq2(float dx, float dy) {
#pragma clang fp contract(off)
  return dx * dx + dy * dy;
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

// ---------------------------------------------------------------------------------------
// max splat.  Two ideas on top of the lock-free packed atomicMax:
//  * dominance pruning: in a dense splat every pixel is hit by tens of points but only the
//    running maximum matters.  A cheap fp32 upper bound of feature*weight (hardware cosine
//    +- 2e-5) is compared with the pixel's CURRENT value, read with an L2-served relaxed
//    atomic load (values only grow, so a stale read can only under-prune);
//  * wave-level compaction: the few surviving hits (~5 %) are appended to a per-wave LDS queue
//    (ballot + mbcnt) and drained 64 at a time, so the fp64 cosine and the 64-bit atomicMax
//    run on full waves instead of on the 1-3 live lanes of a divergent branch.
// ---------------------------------------------------------------------------------------
// XCD-aware work map (speed only; results do not depend on it).  Workgroup g runs on XCD
// g % 8.  The point list is cut in `batch` equal chunks -- in ComputeDepthMaps chunk i is
// exactly image i -- and chunk i is handled by workgroups of XCD i % 8, so the 64-bit image
// of one view (512 KB per 256x256 plane, 16.8 MB per batch of 32) is touched by ONE 4 MB L2
// per plane instead of cycling through all eight.  Returns the point-group id or -1.
__device__ __forceinline__ long xcd_point_group(int lpp, int npoints, int channels, int batch) {
  const int gpb = 256 / lpp;                                   // groups per workgroup
  const long chunk = ((long)npoints + batch - 1) / batch * channels;  // groups per chunk
  const int bpc = (int)((chunk + gpb - 1) / gpb);              // workgroups per chunk
  const int g = blockIdx.x, xcd = g & 7, r = g >> 3;
  const int cidx = (r / bpc) * 8 + xcd;
  if (cidx >= batch) return -1;
  const long local = (long)(r % bpc) * gpb + threadIdx.x / lpp;
  if (local >= chunk) return -1;
  const long gid = (long)cidx * chunk + local;
  return gid < (long)npoints * channels ? gid : -1;
}
inline long xcd_point_blocks(int lpp, int npoints, int channels, int batch) {
  const int gpb = 256 / lpp;
  const long chunk = ((long)npoints + batch - 1) / batch * channels;
  const long bpc = (chunk + gpb - 1) / gpb;
  return bpc * 8 * ((batch + 7) / 8);
}

struct SplatHit {
  unsigned pix;    // pixel index inside the whole [B,C,H,W] image
  float r, f;
  unsigned low;    // 0xFFFFFFFE - point id
};

// largest fp32 s with sqrtf(s) <= radius (sqrtf correctly rounded and monotone), so that
// "s <= s_max" decides exactly what the reference's "sqrt(dx*dx+dy*dy) <= radius" decides
__device__ __forceinline__ float max_sq_inside(float radius) {
#pragma clang fp contract(off)
  float s = radius * radius;
  while (__builtin_sqrtf(s) > radius) s = __uint_as_float(__float_as_uint(s) - 1u);
  for (;;) {
    const float t = __uint_as_float(__float_as_uint(s) + 1u);
    if (!(__builtin_sqrtf(t) <= radius)) break;
    s = t;
  }
  return s;
}

__device__ __forceinline__ void splat_exact(const SplatHit &hh, unsigned long long *img,
                                            float radius) {
  const float v = hh.f * cos_weight(__builtin_sqrtf(hh.r), radius);  // hh.r holds dx*dx+dy*dy
  atomicMax(img + hh.pix, ((unsigned long long)ord_f32(v) << 32) | hh.low);
}

template <int LPP>
__global__ __launch_bounds__(256) void p2i_max_splat_kernel(
    const float *__restrict__ points, const float *__restrict__ feat,
    const int *__restrict__ batch_inds, unsigned long long *__restrict__ img, int npoints,
    int channels, int batch, int h, int w, float radius) {
  constexpr int kQ = 128;
  __shared__ SplatHit queue[4][kQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  SplatHit *q = queue[wave];
  const long gid = xcd_point_group(LPP, npoints, channels, batch);
  const int l = threadIdx.x % LPP;
  bool alive = gid >= 0;
  int x = 0, y = 0, min_y = 0, max_y = -1, max_x = -1;
  float px = 0.f, py = 0.f, f = 0.f;
  unsigned low = 0, plane = 0;
  if (alive) {
    const int c = (int)(gid % channels);
    const int pid = (int)((gid / channels) % npoints);
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) {
      alive = false;
    } else {
      py = points[pid * 2 + 0];
      px = points[pid * 2 + 1];
      f = feat[gid];
      const Box bx = box_of(py, px, h, w, radius);
      x = bx.min_x + l;
      y = min_y = bx.min_y;
      max_y = bx.max_y;
      max_x = bx.max_x;
      low = 0xFFFFFFFEu - (unsigned)pid;
      plane = (unsigned)(((size_t)b * channels + c) * h * w);
      alive = x <= max_x;
    }
  }
  const float rev_scale = 0.5f / radius;  // r*pi/R radians = r/(2R) revolutions
  const float s_max = max_sq_inside(radius);
  int qn = 0;                             // wave-uniform queue fill
  constexpr int kRows = 8;                // rows of one column handled per trip: 8 independent
                                          // L2 reads in flight instead of one dependent read
  while (__any(alive)) {
    float s2[kRows];
    unsigned pix[kRows], cur[kRows];
    bool in[kRows];
    const float dx = x - px;
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      const int yy = y + i;
      s2[i] = sq2(dx, yy - py);
      in[i] = alive && yy <= max_y && s2[i] <= s_max;  // <=> sqrtf(s2) <= radius
      pix[i] = in[i] ? plane + (unsigned)yy * w + x : plane;
    }
#pragma unroll
    for (int i = 0; i < kRows; ++i)
      cur[i] = (unsigned)(__hip_atomic_load(img + pix[i], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT) >> 32);
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      bool pass = false;
      if (in[i]) {
        const float ra = __builtin_amdgcn_sqrtf(s2[i]);  // ~1 ulp: only feeds the bound
        const float wq = __builtin_amdgcn_cosf(ra * rev_scale) * 0.5f + 0.5f;
        const float ub = f >= 0.f ? f * (wq + 2e-5f) : f * __builtin_fmaxf(wq - 2e-5f, 0.f);
        pass = !(ord_f32(ub) < cur[i]);  // can still reach (or tie with) the current value
      }
      const unsigned long long m = __ballot(pass);
      if (m) {
        if (pass)
          q[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
              SplatHit{pix[i], s2[i], f, low};  // exact r is recomputed on the exact path
        qn += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (qn >= 64) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          qn -= 64;
          splat_exact(q[qn + lane], img, radius);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
    }
    if (alive) {
      y += kRows;
      if (y > max_y) {
        y = min_y;
        x += LPP;
        alive = x <= max_x;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < qn) splat_exact(q[lane], img, radius);
}

// ---------------------------------------------------------------------------------------
// Binned gather (the production path for kernel radii <= 16 px; several radii per pass).
// The scatter above pays an L2 round trip per pixel hit (the pruning read), a device-scope
// atomic per survivor, and wastes lanes on the box corners.  The gather turns the loop inside
// out: points are binned once into 8x8-pixel cells (counting sort, one entry per point); one
// WAVE owns an 8x8 tile, lane = pixel, and walks the points of the (2H+1)^2 surrounding cells,
// H = floor(Rmax / 8) + 1.  A cell row of an image is contiguous in the sorted arrays, so the
// candidates arrive as 2H+1 ranges, fetched 64 at a time (one per lane) and broadcast with
// v_readlane.  Every in-range (candidate, pixel) pair gets a cheap fp32 value with a proven error bound; a pixel
// keeps the three largest in registers and only the winner (and a runner-up inside the error band) is evaluated
// with the reference's fp64 cosine at the end -- see p2i_gather_max_kernel.  Equal values still resolve to the
// lowest point id and nothing depends on the visiting order.  All radii of a ComputeDepthMaps call share the
// binning, the fetches and the squared distances; the finished tile is written once as values + ids.
// ---------------------------------------------------------------------------------------
constexpr int kCell = 8;
constexpr int kMaxRadii = 4;

struct RadiiArg {
  float radius[kMaxRadii], s_max[kMaxRadii], rev_scale[kMaxRadii], inv_r2[kMaxRadii];
  float s_max_all;  // largest s_max
  int halo;         // cells to look at on each side
};

// cell of a point = cell of its centre clamped into the image (a point outside the image can
// only reach pixels within R of the border, which the halo covers); -1: cannot reach any pixel
__device__ __forceinline__ int cell_of_point(float py, float px, int h, int w, int cells_x,
                                             int cells_y) {
  if (!(__builtin_fabsf(py) < 1e9f) || !(__builtin_fabsf(px) < 1e9f)) return -1;  // NaN / inf
  const int cy = clampi((int)floorf(py) / kCell, 0, cells_y - 1);
  const int cx = clampi((int)floorf(px) / kCell, 0, cells_x - 1);
  (void)h;
  (void)w;
  return cy * cells_x + cx;
}

__global__ __launch_bounds__(256) void p2i_bin_count_kernel(
    const float *__restrict__ points, const int *__restrict__ batch_inds, int *__restrict__ offs,
    int npoints, int batch, int h, int w, int cells_x, int cells_y, const unsigned *__restrict__ need) {
  if (*need == 0u) return;  // p2i_bin_grouped_kernel has binned everything
  for (int pid = blockIdx.x * blockDim.x + threadIdx.x; pid < npoints;
       pid += gridDim.x * blockDim.x) {
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const int c = cell_of_point(points[pid * 2 + 0], points[pid * 2 + 1], h, w, cells_x, cells_y);
    if (c >= 0) atomicAdd(&offs[b * cells_y * cells_x + c], 1);
  }
}

// exclusive scan of the cell counts, one workgroup per image: the workgroup first adds up the
// counts of all earlier images (its base offset), then scans its own cells.  offs_out may not
// alias counts (other workgroups still read them).
__global__ __launch_bounds__(1024) void p2i_bin_scan_kernel(const int *__restrict__ counts,
                                                            int *__restrict__ offs_out,
                                                            int cells_per_image,
                                                            const unsigned *__restrict__ need) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (*need == 0u) return;
  const int tid = threadIdx.x, b = blockIdx.x;
  int before = 0;
  for (long i = tid; i < (long)b * cells_per_image; i += 1024) before += counts[i];
  for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
  if ((tid & 63) == 0) wsum[tid >> 6] = before;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wsum[w];
    carry = t;
  }
  __syncthreads();
  const int *c = counts + (size_t)b * cells_per_image;
  int *o = offs_out + (size_t)b * cells_per_image;
  for (int base = 0; base < cells_per_image; base += 1024) {
    const int i = base + tid;
    const int v = i < cells_per_image ? c[i] : 0;
    int incl = v;
    for (int m = 1; m < 64; m <<= 1) {
      const int u = __shfl_up(incl, m);
      if ((tid & 63) >= m) incl += u;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int pre = carry;
    for (int wv = 0; wv < (tid >> 6); ++wv) pre += wsum[wv];
    if (i < cells_per_image) o[i] = pre + incl - v;
    __syncthreads();
    if (tid == 1023) carry = pre + incl;
    __syncthreads();
  }
}

// sorted payload: srec[pos] = {row, col, feature (single-channel case), point id bits} -- one
// 16-byte load per candidate in the gather; afterwards offs[c] is the END offset of cell c
__global__ __launch_bounds__(256) void p2i_bin_scatter_kernel(
    const float *__restrict__ points, const float *__restrict__ feat,
    const int *__restrict__ batch_inds, int *__restrict__ offs, float4 *__restrict__ srec,
    unsigned *__restrict__ fmax_bits, int npoints, int channels, int batch, int h, int w, int cells_x,
    int cells_y, const unsigned *__restrict__ need) {
  if (*need == 0u) return;
  float fm = 0.f;  // largest |feature| of the binned points: scales the error bound of the gather's fast path
  for (int pid = blockIdx.x * blockDim.x + threadIdx.x; pid < npoints;
       pid += gridDim.x * blockDim.x) {
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const int c = cell_of_point(py, px, h, w, cells_x, cells_y);
    if (c < 0) continue;
    const int pos = atomicAdd(&offs[b * cells_y * cells_x + c], 1);
    srec[pos] = make_float4(py, px, channels == 1 ? feat[pid] : 0.f, __int_as_float(pid));
    for (int ch = 0; ch < channels; ++ch) fm = __builtin_fmaxf(fm, __builtin_fabsf(feat[(size_t)pid * channels + ch]));
  }
  for (int m = 1; m < 64; m <<= 1) fm = __builtin_fmaxf(fm, __shfl_xor(fm, m));
  // non-negative floats order like their bit patterns; the read first keeps 65k waves off one address
  if ((threadIdx.x & 63) == 0 && __float_as_uint(fm) > *reinterpret_cast<volatile unsigned *>(fmax_bits))
    atomicMax(fmax_bits, __float_as_uint(fm));
}

// The common layout -- image b owns the points [b * ppi, (b + 1) * ppi), which is what ComputeDepthMaps passes --
// binned by ONE workgroup per image with its counters in LDS: count, scan and scatter in one launch without a
// single global atomic (the three generic kernels above: 4.2 M global atomics each way for a sweep of 8 views,
// 0.32 ms; this one reads the points twice).  The layout is VERIFIED, not assumed: a workgroup that meets a
// point of another image, or a point without a cell (non-finite coordinates: the image's base offset would no
// longer be b * ppi), raises `need` and the generic kernels, launched behind it, redo the whole job; with
// `need` clear they return at once.  Workgroup b runs on XCD b % 8, where the gather reads image b.
constexpr int kGroupCells = 8192;  // cells per image the LDS counters hold (a 512 x 1024 image)
__global__ __launch_bounds__(1024) void p2i_bin_grouped_kernel(
    const float *__restrict__ points, const float *__restrict__ feat,
    const int *__restrict__ batch_inds, int *__restrict__ offs, float4 *__restrict__ srec,
    unsigned *__restrict__ fmax_bits, unsigned *__restrict__ need, int ppi, int channels, int h, int w,
    int cells_x, int cells_y) {
  __shared__ int cnt[kGroupCells];
  __shared__ int wsum[16];
  __shared__ int carry, bad;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int cpi = cells_x * cells_y;
  const long base = (long)b * ppi;
  for (int i = tid; i < cpi; i += 1024) cnt[i] = 0;
  if (tid == 0) bad = 0;
  __syncthreads();
  bool wrong = false;
  for (int i = tid; i < ppi; i += 1024) {
    const long pid = base + i;
    const int c = cell_of_point(points[pid * 2 + 0], points[pid * 2 + 1], h, w, cells_x, cells_y);
    wrong |= batch_inds[pid] != b || c < 0;
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  if (wrong) bad = 1;
  __syncthreads();
  if (bad) {  // every thread of the workgroup sees the same value
    if (tid == 0) atomicOr(need, 1u);
    return;
  }
  // exclusive scan of the counters -> first record of every cell (the image starts at b * ppi)
  if (tid == 0) carry = (int)base;
  __syncthreads();
  for (int c0 = 0; c0 < cpi; c0 += 1024) {
    const int i = c0 + tid;
    const int v = i < cpi ? cnt[i] : 0;
    int incl = v;
    for (int m = 1; m < 64; m <<= 1) {
      const int u = __shfl_up(incl, m);
      if ((tid & 63) >= m) incl += u;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int pre = carry;
    for (int wv = 0; wv < (tid >> 6); ++wv) pre += wsum[wv];
    if (i < cpi) cnt[i] = pre + incl - v;
    __syncthreads();
    if (tid == 1023) carry = pre + incl;
    __syncthreads();
  }
  float fm = 0.f;
  for (int i = tid; i < ppi; i += 1024) {
    const long pid = base + i;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const int c = cell_of_point(py, px, h, w, cells_x, cells_y);
    const int pos = atomicAdd(&cnt[c], 1);
    srec[pos] = make_float4(py, px, channels == 1 ? feat[pid] : 0.f, __int_as_float((int)pid));
    for (int ch = 0; ch < channels; ++ch) fm = __builtin_fmaxf(fm, __builtin_fabsf(feat[(size_t)pid * channels + ch]));
  }
  __syncthreads();
  for (int i = tid; i < cpi; i += 1024) offs[(long)b * cpi + i] = cnt[i];  // END offsets, as the scatter leaves them
  for (int m = 1; m < 64; m <<= 1) fm = __builtin_fmaxf(fm, __shfl_xor(fm, m));
  if ((tid & 63) == 0 && __float_as_uint(fm) > *reinterpret_cast<volatile unsigned *>(fmax_bits))
    atomicMax(fmax_bits, __float_as_uint(fm));
}

#ifdef SN_P2I_DIAG  // statistics of the gather (diag build only)
__device__ unsigned long long g_gather_diag[8];
#define GDIAG(...) __VA_ARGS__
#else
#define GDIAG(...)
#endif

// (cos(pi r / R) + 1) / 2 from u = r^2 / R^2 in [0, 1]: a degree-5 polynomial fitted on Chebyshev nodes (within
// 5.0e-7 of the function; the Taylor series needs u^9 for that), Horner in fp32.
// |weight32(u) - exact weight| <= 1.5e-6 = kWeightErr covers: fit + fp32 Horner (6.1e-7 measured over every
// fp32 u of a fine grid, tests/test_p2i.py), the two roundings of u (<= 3e-7: |dw/du| <= pi^2 / 4), and the
// reference rounding r = sqrtf(s) to a float before the cosine (<= 1e-7).
__device__ __forceinline__ float weight32(float u) {
  float p = __builtin_fmaf(u, -0.0102886269f, 0.114825197f);
  p = __builtin_fmaf(u, p, -0.666184545f);
  p = __builtin_fmaf(u, p, 2.02902055f);
  p = __builtin_fmaf(u, p, -2.46737266f);
  return __builtin_fmaf(u, p, 0.999999583f);
}
constexpr float kWeightErr = 1.5e-6f;
// weight32(u) + e with e added to the constant term (one instruction less; the sum differs from the two-step form by
// at most an ulp of 1, far inside the 4e-6 the callers add as a bound's slack)
__device__ __forceinline__ float weight32_plus(float u, float e) {
  float p = __builtin_fmaf(u, -0.0102886269f, 0.114825197f);
  p = __builtin_fmaf(u, p, -0.666184545f);
  p = __builtin_fmaf(u, p, 2.02902055f);
  p = __builtin_fmaf(u, p, -2.46737266f);
  return __builtin_fmaf(u, p, 0.999999583f + e);
}

// sin(pi t) / (pi t), t = r / R, from u = t^2 in [0, 1], the same way (within 2.3e-7).  The slope of the weight is
// d/dr = -(pi^2 / 2 R^2) r slope32(u); used by the backward pass only.
__device__ __forceinline__ float slope32(float u) {
  float p = __builtin_fmaf(u, -0.00193834002f, 0.0257028304f);
  p = __builtin_fmaf(u, p, -0.190524563f);
  p = __builtin_fmaf(u, p, 0.811689675f);
  p = __builtin_fmaf(u, p, -1.64492953f);
  return __builtin_fmaf(u, p, 0.99999994f);
}

__global__ void p2i_series_kernel(const float *__restrict__ u, int n, float *__restrict__ w,
                                  float *__restrict__ sl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    w[i] = weight32(u[i]);
    sl[i] = slope32(u[i]);
  }
}

// Binned gather with a DECIDE-CHEAP / EVALUATE-THE-WINNER split.  The reference's value is
// feature * (float)(cos(r pi / R) / 2 + 1 / 2) with the cosine in double; evaluating that for every pair that
// might raise a pixel's running maximum cost 27 % of the renderer (3.2 evaluations per pixel and radius: the
// record-breaking count of a streaming maximum).  Now every in-range pair gets the fp32 series value `a`, which is
// within EPS = max|feature| * kWeightErr of the exact one, and a pixel keeps the TOP THREE of them (value + sorted
// position for the first two).  At the end only the winner is evaluated exactly -- and the runner-up too when it
// lies within 2 EPS of the winner (then the exact values and the lowest point id decide, as before).  If even the
// third lies within the band (equal points in triplicate ...) the pixel is settled by an exact walk over all its
// candidates.  The background takes part as a candidate with an exact value and no id.  Why this is exact: the
// true winner T has a_T >= a_X - 2 EPS for every X, so it is never pruned (a candidate is dropped only when
// a < current best - 2 EPS), and it can only leave the top two if two others lie within the band above it --
// which puts the third inside the band and triggers the exact walk.
template <int NR, bool C1>
__global__ __launch_bounds__(256, 6) void p2i_gather_max_kernel(
    const float *__restrict__ feat, const float *__restrict__ background,
    const float4 *__restrict__ srec, const int *__restrict__ offs, const unsigned *__restrict__ fmax_bits,
    int channels, int batch, int h, int w, int cells_x, int cells_y, RadiiArg ra,
    float *__restrict__ out, int *__restrict__ out_ids, long obstride, long orstride) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  long tile = (long)blockIdx.x * 4 + wave;  // (b, c, cy, cx), cx fastest
  const long tiles = (long)batch * channels * cells_y * cells_x;
  if (tile >= tiles) return;  // whole wave; no workgroup barrier below
  const int cx = (int)(tile % cells_x); tile /= cells_x;
  const int cy = (int)(tile % cells_y); tile /= cells_y;
  const int c = (int)(tile % channels);
  const int b = (int)(tile / channels);
  const int x = cx * kCell + (lane & 7), y = cy * kCell + (lane >> 3);
  const bool valid = x < w && y < h;
  const unsigned long long valid_mask = __builtin_amdgcn_ballot_w64(valid);
  const size_t plane = ((size_t)b * channels + c) * h * w;
  const float fx = (float)x, fy = (float)y;
  const float eps = __uint_as_float(*fmax_bits) * kWeightErr, band = 2.f * eps;
  constexpr unsigned kBg = 0xFFFFFFFFu;  // "position" of the background
  const float bg = valid && background ? background[plane + (size_t)y * w + x] : 0.f;  // nullptr: all zeros
  float b1[NR], b2[NR], b3[NR];   // the three largest fast values seen (b1: also the pruning bound)
  float low1[NR];                 // b1 - band, kept next to b1
  unsigned j1[NR], j2[NR];        // sorted positions (srec index) of the first two
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    b1[k] = bg;
    low1[k] = bg - band;
    j1[k] = kBg;
    b2[k] = b3[k] = -3.0e38f;
    j2[k] = kBg;
  }
  const float tx0 = (float)(cx * kCell), tx1 = tx0 + (kCell - 1), ty0 = (float)(cy * kCell),
              ty1 = ty0 + (kCell - 1);
  float tile_min[NR];  // wave-uniform: smallest b1 over the tile's pixels
  GDIAG(int dg_batches = 0, dg_cand = 0, dg_surv = 0, dg_pairs = 0, dg_upd = 0, dg_amb = 0, dg_walk = 0, dg_rings = 0;)
  bool dirty[NR];  // wave-uniform: some pixel's top three of radius k changed since tile_min[k] was taken
  auto refresh_tile_min = [&]() {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (!dirty[k]) continue;
      dirty[k] = false;
      // float minimum over the wave (wave_dpp.hpp: six v_min_f32_dpp + one v_readlane).  The values are finite;
      // pixels outside the image take part as +3e38.
      tile_min[k] = sn::wave_min_f32(valid ? b1[k] : 3.0e38f);
    }
  };
#pragma unroll
  for (int k = 0; k < NR; ++k) dirty[k] = true;
  refresh_tile_min();

  const int cell_base = b * cells_y * cells_x;
  // the candidate ranges of the tile: rows nearest first, the own row as own cell / left / right
  auto row_range = [&](int step, int part, int &beg, int &end) -> bool {
    const int dyc = (step + 1) >> 1;  // row distance in cells
    const int cyy = cy + ((step & 1) ? -dyc : dyc);
    if (cyy < 0 || cyy >= cells_y) return false;
    // Cells of this row that the largest radius can reach from the tile: a cell dxc columns and dyc rows away is at
    // least ((dxc - 1)+, (dyc - 1)+) * kCell pixels from every pixel of the tile (its points lie inside it, or --
    // border cells -- beyond it), so the corner cells of the (2 halo + 1)^2 block drop out (4 of 25 at R = 10).
    const float gy = (float)((dyc > 1 ? dyc - 1 : 0) * kCell);
    const float rem = ra.s_max_all * 1.0001f - gy * gy;
    if (rem < 0.f) return false;
    int reach = (int)(__builtin_sqrtf(rem) * (1.0f / kCell)) + 1;
    reach = reach < ra.halo ? reach : ra.halo;
    const int c_lo = cx - reach > 0 ? cx - reach : 0;
    const int c_hi = cx + reach < cells_x - 1 ? cx + reach : cells_x - 1;
    const int p_lo = step == 0 ? (part == 0 ? cx : (part == 1 ? c_lo : cx + 1)) : c_lo;
    const int p_hi = step == 0 ? (part == 0 ? cx : (part == 1 ? cx - 1 : c_hi)) : c_hi;
    if (p_lo > p_hi) return false;
    const int first_cell = cell_base + cyy * cells_x + p_lo;
    beg = first_cell > 0 ? offs[first_cell - 1] : 0;
    end = offs[cell_base + cyy * cells_x + p_hi];
    return beg < end;
  };

  // The same cells RING by ring (round 6): ring r = the cells max(|dxc|, |dyc|) = r away -- its top and bottom rows
  // as runs, the two side cells of every row in between.  Every point of ring r >= 2 lies at least (r - 1) kCell
  // pixels from every pixel of the tile, so ONE uniform test per ring and radius -- the largest feature of the call
  // times the weight at that distance against the smallest running best of the tile -- says whether any candidate of
  // the ring (and of every ring beyond it: farther, and the running bests only rise) can still come within the band
  // of some pixel's best.  It is the per-candidate cull's own test with the ring's bounds in place of the candidate's,
  // so what it skips the cull would have dropped one batch of 64 at a time: on the benchmark's clouds ring 2 holds
  // 12 of the 21 cells of a tile at R = 10 (57 % of the candidates, 0.095 of the weight at best) and is skipped for
  // nearly every tile.
  auto ring_range = [&](int ring, int rstep, int part, int &beg, int &end) -> bool {
    const int dyc = (rstep + 1) >> 1;  // row distance in cells (<= ring)
    const int cyy = cy + ((rstep & 1) ? -dyc : dyc);
    if (cyy < 0 || cyy >= cells_y) return false;
    const float gy = (float)((dyc > 1 ? dyc - 1 : 0) * kCell);
    const float rem = ra.s_max_all * 1.0001f - gy * gy;
    if (rem < 0.f) return false;
    int p_lo, p_hi;
    if (dyc == ring) {  // the ring's top / bottom row (ring 0: the own cell): one run of cells
      // (Measured and not kept: the 3 x 3 block as three runs of three cells -- fuller batches for the cull, 3.9 instead
      // of 5.3 per tile -- lets the neighbours arrive before the own cell has raised the tile's bests: 51 survivors per
      // tile instead of 35, gather 1.27 -> 1.50 ms, profiles/r06_i_render_kernels_3x3_runs_not_kept.txt.)
      if (part != 0) return false;
      int reach = (int)(__builtin_sqrtf(rem) * (1.0f / kCell)) + 1;
      reach = reach < ring ? reach : ring;
      p_lo = cx - reach > 0 ? cx - reach : 0;
      p_hi = cx + reach < cells_x - 1 ? cx + reach : cells_x - 1;
    } else {  // an inner row: the ring's left (part 0) and right (part 1) cell
      const float gx = (float)((ring - 1) * kCell);
      if (gx * gx > rem) return false;  // the corner cells the largest radius cannot reach
      p_lo = p_hi = part == 0 ? cx - ring : cx + ring;
      if (p_lo < 0 || p_lo >= cells_x) return false;
    }
    const int first_cell = cell_base + cyy * cells_x + p_lo;
    beg = first_cell > 0 ? offs[first_cell - 1] : 0;
    end = offs[cell_base + cyy * cells_x + p_hi];
    return beg < end;
  };
  const float fmaxv = __uint_as_float(*fmax_bits);
#ifndef SN_P2I_NO_RINGS
  for (int ring = 0; ring <= ra.halo; ++ring) {
    if (ring >= 2) {  // can anything of this ring still matter?  (uniform values: one answer for the wave)
      const float g = (float)((ring - 1) * kCell);
      const float g2 = g * g * 0.99999f;
      bool live = false;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const float ub = fmaxv * weight32_plus(__builtin_fminf(g2 * ra.inv_r2[k], 1.0f), 4e-6f);
        live = live || (g2 <= ra.s_max[k] && tile_min[k] <= ub + band);
      }
      GDIAG(dg_rings += live ? 0 : 1;)
      if (!live) break;
    }
   for (int rstep = 0; rstep <= 2 * ring; ++rstep) {
    for (int part = 0; part < 2; ++part) {
      int beg, end;
      if (!ring_range(ring, rstep, part, beg, end)) continue;  // wave-uniform
#else
  {
   for (int step = 0; step <= 2 * ra.halo; ++step) {
    const int nparts = step == 0 ? 3 : 1;
    for (int part = 0; part < nparts; ++part) {
      int beg, end;
      if (!row_range(step, part, beg, end)) continue;  // wave-uniform
#endif
      float4 rec_next = make_float4(0.f, 0.f, 0.f, 0.f);
      if (beg + lane < end) rec_next = srec[beg + lane];
      for (int base = beg; base < end; base += 64) {
        const int j = base + lane;
        const float4 rec = rec_next;
        if (j + 64 < end) rec_next = srec[j + 64];  // the next 64 candidates are in flight
        float cpy = 0.f, cpx = 0.f, cf = 0.f;
        if (j < end) {
          cpy = rec.x;
          cpx = rec.y;
          cf = C1 ? rec.z : feat[(size_t)__float_as_int(rec.w) * channels + c];
        }
        // Cull, 64 candidates at a time (lane = candidate): nearest pixel of the tile out of range, or even
        // there its value cannot come within the band of the smallest running best of the tile.
        unsigned long long keep[NR];
        {
          const float ddx = __builtin_fmaxf(__builtin_fmaxf(tx0 - cpx, cpx - tx1), 0.f);
          const float ddy = __builtin_fmaxf(__builtin_fmaxf(ty0 - cpy, cpy - ty1), 0.f);
          const float smin = (ddx * ddx + ddy * ddy) * 0.99999f;  // <= every pixel's s, with slack
          // lane masks straight from the compares (see the survivor loop); max(cf, 0): weights are >= 0
          unsigned long long m_end;
          float cfp;
          asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m_end) : "v"(j), "s"(end));
          asm("v_max_f32 %0, %1, %2" : "=v"(cfp) : "v"(cf), "v"(0.f));
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            // weight32 decreases in u: its value at the nearest pixel (+ its error, folded into the constant term)
            // bounds the candidate's weights
            float u;
            asm("v_min_f32 %0, %1, %2" : "=v"(u) : "v"(smin * ra.inv_r2[k]), "v"(1.0f));
            const float ub = cfp * weight32_plus(u, 4e-6f);
            unsigned long long m_r, m_v;
            asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m_r) : "s"(ra.s_max[k]), "v"(smin));
            asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m_v) : "s"(tile_min[k]), "v"(ub + band));
            keep[k] = m_end & m_r & m_v;
          }
        }
        unsigned long long todo = keep[0];
#pragma unroll
        for (int k = 1; k < NR; ++k) todo |= keep[k];
        GDIAG(dg_batches++; dg_cand += (end - base < 64 ? end - base : 64); dg_surv += __popcll(todo);)
        while (todo) {  // wave-uniform: candidate i broadcast to every pixel
          const int i = __builtin_ctzll(todo);
          todo &= todo - 1;
          const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cpy), i));
          const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cpx), i));
          const float f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), i));
          const float dx = fx - px, dy = fy - py;
          const float s2 = sq2(dx, dy);
          const unsigned pos = (unsigned)(base + i);
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            if (!((keep[k] >> i) & 1ull)) continue;  // wave-uniform
            const float a = f * weight32(s2 * ra.inv_r2[k]);  // out of range: a finite value nobody looks at
            // Lane masks straight from the compares (inline assembly: through `bool`s hipcc materialises a 0 / 1
            // vector and compares it again for the wave-level test): in range <=> s2 <= s_max[k] <=> sqrtf(s2) <=
            // radius[k]; candidate for the top three <=> a >= b1 - band.
            unsigned long long m_in, m_ge;
            asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m_in) : "s"(ra.s_max[k]), "v"(s2));
            asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m_ge) : "v"(a), "v"(low1[k]));
            const unsigned long long pass = m_in & m_ge & valid_mask;
            GDIAG(dg_pairs += __popcll(m_in & valid_mask); dg_upd += __popcll(pass);)
            if (pass != 0ull) {  // scalar test and branch
              dirty[k] = true;
              // Sorted insertion of `a` into b1 >= b2 >= b3 on EVERY lane, as a max / min ladder: lanes outside `pass`
              // insert -inf, which leaves all three where they are and never wins a (strict) position compare -- also
              // against a background of -inf.  (A NaN background turns into -inf here; such a pixel never passes
              // and the epilogue compares against the background itself, so it still comes out as the background.)
              // Equal values keep their order of arrival, exactly like the compare chain of the other branch.
              // (inline assembly: hipcc puts a canonicalising v_max(x, x) in front of every fmaxf / fminf on a
              // loop-carried value; a, b1..b3 are finite -- products of finite numbers -- never NaN)
              // (Round 6 measured a TWO-value ladder -- 8 instructions instead of 13, every pixel with a runner-up inside
              // the band settled by the exact walk instead of two exact evaluations: 0.022 pixel slots per tile walk, and
              // the gather goes from 1.27 to 1.30 ms: profiles/r06_j_render_kernels_two_value_ladder_not_kept.txt.)
              float ae, t, t2, n1, n2, n3;
              unsigned long long first, second;  // lane masks of ae > b1, ae > b2
              unsigned posv = pos, e2, nj1, nj2;
              const float far = -__builtin_inff();
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(ae) : "v"(far), "v"(a), "s"(pass));
              asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(first) : "v"(ae), "v"(b1[k]));
              asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(second) : "v"(ae), "v"(b2[k]));
              asm("v_min_f32 %0, %1, %2" : "=v"(t) : "v"(b1[k]), "v"(ae));
              asm("v_max_f32 %0, %1, %2" : "=v"(n1) : "v"(b1[k]), "v"(ae));
              asm("v_min_f32 %0, %1, %2" : "=v"(t2) : "v"(b2[k]), "v"(t));
              asm("v_max_f32 %0, %1, %2" : "=v"(n2) : "v"(b2[k]), "v"(t));
              asm("v_max_f32 %0, %1, %2" : "=v"(n3) : "v"(b3[k]), "v"(t2));
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(e2) : "v"(j2[k]), "v"(posv), "s"(second));
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nj2) : "v"(e2), "v"(j1[k]), "s"(first));
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nj1) : "v"(j1[k]), "v"(posv), "s"(first));
              j2[k] = nj2;
              j1[k] = nj1;
              b1[k] = n1;
              b2[k] = n2;
              b3[k] = n3;
              low1[k] = n1 - band;
            }
          }
        }
        refresh_tile_min();
      }
    }
   }
  }

  // ---- exact values: the winner, the runner-up when it is inside the band, an exact walk when the third is too
  auto exact_of = [&](unsigned pos, double inv_radius, float s_max, float &v, unsigned &low) {
    // value and tie key (0xFFFFFFFE - id: larger = lower id) of the candidate at sorted position `pos`
    const float4 rec = srec[pos];
    const int pid = __float_as_int(rec.w);
    const float f = C1 ? rec.z : feat[(size_t)pid * channels + c];
    const float s2 = sq2(fx - rec.y, fy - rec.x);
    (void)s_max;
    v = f * cos_weight_inv(__builtin_sqrtf(s2), inv_radius);
    low = 0xFFFFFFFEu - (unsigned)pid;
  };
  // output position: image b, radius k, channel c -> b * obstride + k * orstride + c * h * w (+ pixel): radius-major
  // [R,B,C,H,W] or image-major [B,R,C,H,W], chosen by the caller
  const size_t oplane = (size_t)b * obstride + (size_t)c * h * w;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    float best_v = bg;          // the reference replaces only on strictly greater: the background wins ties
    unsigned best_low = kBg;    // kBg = no point
    const double inv_rd = 1.0 / (double)ra.radius[k];  // one value for every exact evaluation of this radius
    const bool walk = valid && b3[k] >= b1[k] - band;
    const bool amb = valid && !walk && b2[k] >= b1[k] - band;
    GDIAG(dg_amb += __popcll(__ballot(amb)); dg_walk += __popcll(__ballot(walk));)
    if (valid && !walk) {
      auto consider = [&](unsigned pos) {
        if (pos == kBg) return;
        float v;
        unsigned low;
        exact_of(pos, inv_rd, ra.s_max[k], v, low);
        if (v > best_v || (v == best_v && best_low != kBg && low > best_low)) {
          best_v = v;
          best_low = low;
        }
      };
      consider(j1[k]);
      if (amb) consider(j2[k]);
    }
    if (__any(walk)) {  // rare: three or more candidates inside the band -- settle the pixel exactly
      for (int step = 0; step <= 2 * ra.halo; ++step) {
        const int nparts = step == 0 ? 3 : 1;
        for (int part = 0; part < nparts; ++part) {
          int beg, end;
          if (!row_range(step, part, beg, end)) continue;
          for (int pos = beg; pos < end; ++pos) {
            const float4 rec = srec[pos];  // wave-uniform address
            const float s2 = sq2(fx - rec.y, fy - rec.x);
            if (walk && s2 <= ra.s_max[k]) {
              const int pid = __float_as_int(rec.w);
              const float f = C1 ? rec.z : feat[(size_t)pid * channels + c];
              const float v = f * cos_weight_inv(__builtin_sqrtf(s2), inv_rd);
              const unsigned low = 0xFFFFFFFEu - (unsigned)pid;
              if (v > best_v || (v == best_v && best_low != kBg && low > best_low)) {
                best_v = v;
                best_low = low;
              }
            }
          }
        }
      }
    }
    if (valid) {
      const size_t e = (size_t)k * orstride + oplane + (size_t)y * w + x;
      out[e] = best_v;
      out_ids[e] = best_low == kBg ? -1 : (int)(0xFFFFFFFEu - best_low);
    }
  }
  GDIAG(if (lane == 0) {
    atomicAdd(&g_gather_diag[0], 1ull); atomicAdd(&g_gather_diag[1], (unsigned long long)dg_batches);
    atomicAdd(&g_gather_diag[2], (unsigned long long)dg_cand); atomicAdd(&g_gather_diag[3], (unsigned long long)dg_surv);
    atomicAdd(&g_gather_diag[4], (unsigned long long)dg_pairs); atomicAdd(&g_gather_diag[5], (unsigned long long)dg_upd);
    atomicAdd(&g_gather_diag[6], (unsigned long long)dg_amb); atomicAdd(&g_gather_diag[7], (unsigned long long)dg_walk + ((unsigned long long)dg_rings << 32));  // rings skipped: high half
  })
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

// ---------------------------------------------------------------------------------------
// max backward in two dense passes, no atomics.
// The reference walks the pixels and scatters into the winning point with three fp32 atomics
// per pixel (device-scope atomics leave the XCD on MI355X and serialise on popular points).
//   pass 1 (per pixel, fully dense): the fp64 sine/cosine terms of every pixel that has a
//           winner -> three per-pixel contribution planes (d feature, d row, d col);
//           background_grad on the side.
//   pass 2 (per point, gather): a point walks its own footprint (the forward's pixel walk,
//           LPP lanes per point), sums the contributions of the pixels it won
//           (out_ids == point id) in a fixed order and writes its gradient once.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void p2i_max_bwd_pixels_kernel(
    const float *__restrict__ out_grad, const int *__restrict__ out_ids,
    const float *__restrict__ points, const float *__restrict__ feat,
    float *__restrict__ background_grad, float *__restrict__ contrib, int channels, int h, int w,
    float radius, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const float g = out_grad[e];
    const int pid = out_ids[e];
    float cf = 0.f, cy = 0.f, cx = 0.f;
    if (pid >= 0) {
      const int x = (int)(e % w), y = (int)((e / w) % h);
      const int c = (int)((e / ((long)w * h)) % channels);
      const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
      const float dx = x - px, dy = y - py;
      const float r = radius_of(dx, dy);
      const float wgt = cos_weight(r, radius);
      const float fv = feat[(size_t)pid * channels + c];
      cf = g * wgt;
      const float wg = g * fv;
      const float rm = r > 1e-10f ? r : 1e-10f;
      const float k = (float)((double)wg * sin((double)r * M_PI / (double)radius) * 0.5 * M_PI /
                              (double)radius / (double)rm);
      cy = k * dy;
      cx = k * dx;
    }
    background_grad[e] = pid < 0 ? g : 0.f;
    contrib[e] = cf;
    contrib[total + e] = cy;
    contrib[2 * total + e] = cx;
  }
}

template <int LPP>
__global__ __launch_bounds__(256) void p2i_max_bwd_points_kernel(
    const int *__restrict__ out_ids, const float *__restrict__ contrib,
    const float *__restrict__ points, const int *__restrict__ batch_inds,
    float *__restrict__ points_grad, float *__restrict__ feat_grad, int npoints, int channels,
    int batch, int h, int w, float radius, long total) {
  // thread group = one point; channels are walked inside (points_grad sums over channels)
  const long gid = xcd_point_group(LPP, npoints, 1, batch);
  const int l = threadIdx.x % LPP;
  if (gid < 0) return;
  const int pid = (int)gid;
  const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
  const Box bx = box_of(py, px, h, w, radius);
  const float s_max = max_sq_inside(radius);
  // with batch_inds the point's image is known; without (the reference's backward does not
  // receive it) every image plane is searched -- a point id is unique across the batch
  const int bi = batch_inds ? batch_inds[pid] : -1;
  const int b_lo = batch_inds ? bi : 0, b_hi = batch_inds ? bi + 1 : batch;
  float gy = 0.f, gx = 0.f;
  for (int c = 0; c < channels; ++c) {
    float gf = 0.f;
    for (int b = b_lo; b < b_hi; ++b) {
      if (b < 0 || b >= batch) continue;
      const size_t plane = ((size_t)b * channels + c) * h * w;
      for (int x = bx.min_x + l; x <= bx.max_x; x += LPP) {
        const float dx = x - px;
        for (int y0 = bx.min_y; y0 <= bx.max_y; y0 += 8) {
          int ids[8];
          bool in[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {  // 8 independent id reads in flight
            const int yy = y0 + i;
            in[i] = yy <= bx.max_y && sq2(dx, yy - py) <= s_max;
            ids[i] = out_ids[plane + (size_t)(in[i] ? yy : y0) * w + x];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (!in[i] || ids[i] != pid) continue;
            const size_t e = plane + (size_t)(y0 + i) * w + x;
            gf += contrib[e];
            gy += contrib[total + e];
            gx += contrib[2 * total + e];
          }
        }
      }
    }
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) gf += __shfl_xor(gf, m, LPP);
    if (l == 0) feat_grad[(size_t)pid * channels + c] = gf;
  }
#pragma unroll
  for (int m = 1; m < LPP; m <<= 1) {
    gy += __shfl_xor(gy, m, LPP);
    gx += __shfl_xor(gx, m, LPP);
  }
  if (l == 0) {
    points_grad[pid * 2 + 0] = gy;
    points_grad[pid * 2 + 1] = gx;
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

// lanes that share one point's footprint (box at most ceil(2R)+2 pixels wide; a lane walks
// columns l, l+LPP, ...).  Forward: one pass, LPP = 2^k >= width -- the lanes of a point then
// hit consecutive pixels of a row with their loads / atomics.  Backward gather: 8 lanes and
// several passes waste the fewest lanes (R=10: 22 columns = 3 passes of 8, 92 % busy); measured
// 0.14 ms vs 0.19 ms per call, while the forward is 10 % slower with 8.
int lanes_per_point(float radius) {
  const int width = (int)ceilf(2.f * (radius > 0.f ? radius : 0.f)) + 2;
  int l = 4;
  while (l < width && l < 64) l *= 2;
  return l;
}
int lanes_per_point_gather(float radius) {
  const int width = (int)ceilf(2.f * (radius > 0.f ? radius : 0.f)) + 2;
  if (width <= 4) return 4;
  if (width <= 48) return 8;
  if (width <= 128) return 16;
  return 64;
}

int check_common(const char *fn, int npoints, int channels, int batch, int h, int w,
                 float radius) {
  if (!(npoints >= 0 && channels >= 1 && batch >= 1 && h >= 1 && w >= 1))
    return sn::fail(SN_EINVAL, "%s: bad sizes npoints=%d channels=%d batch=%d h=%d w=%d", fn,
                    npoints, channels, batch, h, w);
  if (!(radius > 0.f)) return sn::fail(SN_EINVAL, "%s: kernel radius must be > 0", fn);
  if ((long)batch * channels * h * w >= (1L << 31) - 1 || (long)npoints * channels >= (1L << 31))
    return sn::fail(SN_EINVAL, "%s: tensor too large", fn);
  return 0;
}

int lin_blocks(long total) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------------------------------
// max backward, pixel-centric with EXACT accumulation (all radii of a call in one pass).
// Every pixel that has a winner adds its three terms (d feature, d row, d col) to the winner's
// accumulators.  fp32 atomics would make the sums depend on the arrival order; instead the
// terms are converted to 64-bit fixed point (scale = a power of two chosen on the device from
// max|out_grad| and max|feature|) and added with integer atomics -- integer addition is
// associative, so the result is bit-reproducible, and it equals the exactly rounded sum of the
// fp32 terms to ~2^-45 of the largest possible term.  The XCD-aware map keeps all pixels of an
// image, hence all atomics on its points, inside one XCD's L2.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void p2i_absmax_kernel(const float *__restrict__ a, long na,
                                                         const float *__restrict__ b, long nb,
                                                         unsigned *__restrict__ out2) {
  float ma = 0.f, mb = 0.f;
  // 16-byte loads where the array allows it (torch tensors do): this pass streams 200 MB of gradients at C3
  const long na4 = (reinterpret_cast<size_t>(a) & 15) == 0 ? na / 4 : 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < na4; e += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4 *>(a)[e];
    ma = __builtin_fmaxf(__builtin_fmaxf(ma, __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))),
                         __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
  }
  for (long e = na4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; e < na; e += (long)gridDim.x * blockDim.x)
    ma = __builtin_fmaxf(ma, __builtin_fabsf(a[e]));
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nb; e += (long)gridDim.x * blockDim.x)
    mb = __builtin_fmaxf(mb, __builtin_fabsf(b[e]));
  for (int m = 1; m < 64; m <<= 1) {
    ma = __builtin_fmaxf(ma, __shfl_xor(ma, m));
    mb = __builtin_fmaxf(mb, __shfl_xor(mb, m));
  }
  __shared__ float red[2][16];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = ma;
    red[1][threadIdx.x >> 6] = mb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // |x| >= 0: the bit patterns order like the values
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      ma = __builtin_fmaxf(ma, red[0][i]);
      mb = __builtin_fmaxf(mb, red[1][i]);
    }
    // same-address atomics serialise (~25 ns each): only a block that would raise the maximum issues one
    if (__float_as_uint(ma) > *reinterpret_cast<volatile unsigned *>(out2 + 0)) atomicMax(out2 + 0, __float_as_uint(ma));
    if (__float_as_uint(mb) > *reinterpret_cast<volatile unsigned *>(out2 + 1)) atomicMax(out2 + 1, __float_as_uint(mb));
  }
}

// 2^e with every |term| * 2^e < 2^44 (sums of < 2^18 terms stay inside int64)
__device__ __forceinline__ double fixed_scale(const unsigned *absmax, float min_radius) {
  const float g = __uint_as_float(absmax[0]), f = __uint_as_float(absmax[1]);
  const float m = g * __builtin_fmaxf(1.f, f * (1.5707964f / min_radius) * 1.01f);
  if (!(m > 0.f) || !(m < 3e38f)) return 1.0;
  return ldexp(1.0, 43 - ilogbf(m));
}

#ifndef SN_ACC_SLOTS
#define SN_ACC_SLOTS 512
#endif
#ifndef SN_ACC_REGION
#define SN_ACC_REGION 4   // tiles per side of a workgroup's region (4: 32 x 32 pixels)
#endif
constexpr int kAccSlots = SN_ACC_SLOTS;  // per-WORKGROUP hash table of the winners of a region (real regions of 32 x 32
                                         // pixels hold ~120 distinct winners over three radii; a term that finds no slot
                                         // within kAccProbes steps goes straight to the global accumulators -- integer
                                         // sums are exact in any order): 14 KB of LDS, 8 workgroups per CU
constexpr int kAccProbes = 8;
constexpr int kAccRegion = SN_ACC_REGION;

// One WORKGROUP per region of kAccRegion x kAccRegion tiles of 8 x 8 pixels (wave w walks down column w of the region:
// at any time the four waves read four horizontally adjacent tiles), one hash table per workgroup, ONE flush per
// region.  Round 6, from the experiment matrix of profiles/r06_a_render_accum_experiments.txt (8 views x 32 clouds x 3
// radii: the kernel's 0.49 ms = 0.20 global flush + 0.11 LDS adds + 0.07 dependent gathers + 0.11 streaming): a point
// wins pixels in 2.8 tiles of 8 x 8 on average, so a table per tile (rounds 1-5: 128 slots per wave) sent 2.8 triples
// of 64-bit global atomics per winning point (11.8 M atomics per launch); a region of 32 x 32 pixels holds most
// winners' whole footprint.
__global__ __launch_bounds__(256) void p2i_max_bwd_accum_kernel(
    const float *__restrict__ out_grad, const int *__restrict__ out_ids,
    const float *__restrict__ points, const float *__restrict__ feat,
    const unsigned *__restrict__ absmax, float *__restrict__ background_grad,
    long long *__restrict__ acc_pts, long long *__restrict__ acc_feat, int channels, int batch,
    int h, int w, RadiiArg ra, int nradii, float min_radius, long obstride, long orstride) {
  // lane = pixel of an 8 x 8 tile.  Neighbouring pixels, and the radii of one pixel, often share their winner: the
  // terms first meet in the workgroup's LDS hash table keyed by the point id (LDS integer atomics), and every distinct
  // winner of the region then costs three global atomics.  XCD-aware: workgroup g runs on XCD g % 8 and only touches
  // images b with b % 8 == g % 8.
  __shared__ int keys[kAccSlots];
  __shared__ unsigned long long vals[kAccSlots][3];
  const int cells_x = (w + kCell - 1) / kCell, cells_y = (h + kCell - 1) / kCell;
  const int regs_x = (cells_x + kAccRegion - 1) / kAccRegion, regs_y = (cells_y + kAccRegion - 1) / kAccRegion;
  const int rpi = channels * regs_y * regs_x;                  // regions (= workgroups) per image
  const int g = blockIdx.x, xcd = g & 7, r = g >> 3;
  const int b = (r / rpi) * 8 + xcd;
  if (b >= batch) return;  // whole workgroups
  int reg = r % rpi;
  const int rx = reg % regs_x; reg /= regs_x;
  const int ry = reg % regs_y;
  const int c = reg / regs_y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double scale = fixed_scale(absmax, min_radius);
  const size_t plane = ((size_t)b * channels + c) * h * w;
  for (int i = threadIdx.x; i < kAccSlots; i += 256) {
    keys[i] = -1;
    vals[i][0] = vals[i][1] = vals[i][2] = 0ull;
  }
  __syncthreads();
  // llrint(x * scale) through the 1.5 * 2^52 trick: scale is a power of two (the product is exact) and every
  // |term| < 2^44, so rn(x * scale + 1.5 * 2^52) carries rint(x * scale) in its low mantissa bits -- the same
  // integer as llrint under round-to-nearest-even, in 4 instructions instead of the library's 8 fp64 ones per term
  auto fixed = [&](float x) {
    const double m = 6755399441055744.0;  // 1.5 * 2^52
    return (unsigned long long)(__double_as_longlong(__builtin_fma((double)x, scale, m)) - __double_as_longlong(m));
  };
  for (int ty = 0; ty < kAccRegion; ++ty) {
    for (int tx = wave; tx < kAccRegion; tx += 4) {
      const int cx = rx * kAccRegion + tx, cy = ry * kAccRegion + ty;
      if (cx >= cells_x || cy >= cells_y) continue;  // wave-uniform
      // (Measured and not kept, round 6: a wave's 64 pixels as an 8 x 8 lattice of spacing 4 over the region, so that
      // same-winner neighbours land in different LDS instructions: render 2.03 -> 2.15 ms, the strided reads cost more
      // than the conflicts: profiles/r06_p_accum_lattice_not_kept.txt.)
      const int x = cx * kCell + (lane & 7), y = cy * kCell + (lane >> 3);
      const bool valid = x < w && y < h;
      const size_t e = plane + (size_t)(valid ? y : 0) * w + (valid ? x : 0);
      const size_t oe = (size_t)b * obstride + (size_t)c * h * w + (size_t)(valid ? y : 0) * w + (valid ? x : 0);
      float bg = 0.f;
      float gk[kMaxRadii];
      int pidk[kMaxRadii];
#pragma unroll
      for (int k = 0; k < kMaxRadii; ++k) {  // the tile's loads go out together
        gk[k] = (k < nradii && valid) ? out_grad[(size_t)k * orstride + oe] : 0.f;
        pidk[k] = (k < nradii && valid) ? out_ids[(size_t)k * orstride + oe] : -1;
      }
      // The radii of a pixel often share their winner (the same point wins at 5, 7 and 10 pixels): consecutive equal
      // winners are summed in registers and meet the table once (round 6; integer sums are exact in any order).
      int ppid = -1;
      unsigned long long p0 = 0ull, p1 = 0ull, p2 = 0ull;
#pragma unroll
      for (int k = 0; k <= kMaxRadii; ++k) {
        if (k > nradii) break;
        int pid = -1;
        unsigned long long t0 = 0ull, t1 = 0ull, t2 = 0ull;
        if (k < kMaxRadii && k < nradii) {
          pid = pidk[k];
          if (pid < 0) {
            bg += gk[k];
          } else {
            // weight and slope as fp32 series in u = r^2 / R^2 (no square root, no division, no fp64): within 5e-7 of
            // the reference's fp32 cos / sin; the kernel was bound by the fp64 sincos of every term
            const float2 pp = reinterpret_cast<const float2 *>(points)[pid];  // {row, column}
            const float fv = feat[(size_t)pid * channels + c];
            const float dx = x - pp.y, dy = y - pp.x;
            const float u = __builtin_fminf((dx * dx + dy * dy) * ra.inv_r2[k], 1.0f);
            const float cf = gk[k] * weight32(u);
            const float kk = gk[k] * fv * (4.93480220f * ra.inv_r2[k]) * slope32(u);  // pi^2 / 2
            t0 = fixed(cf);
            t1 = fixed(kk * dy);
            t2 = fixed(kk * dx);
          }
        }
        if (pid >= 0 && pid == ppid) {  // the same winner as the previous radius: one table entry for both
          p0 += t0;
          p1 += t1;
          p2 += t2;
          continue;
        }
        if (ppid >= 0) {
          // (Round 5 merged the terms of neighbouring pixels with the same winner -- lane ^ 1, then lane ^ 8 -- before
          // the table's atomics: exact, and no faster; removed.)
          unsigned slot = (((unsigned)ppid * 2654435761u) >> 16) & (kAccSlots - 1);
          bool found = false;
          for (int probe = 0; probe < kAccProbes; ++probe) {
            const int prev = atomicCAS(&keys[slot], -1, ppid);
            if (prev == -1 || prev == ppid) {
              found = true;
              break;
            }
            slot = (slot + 1) & (kAccSlots - 1);
          }
          if (found) {
            atomicAdd(&vals[slot][0], p0);
            atomicAdd(&vals[slot][1], p1);
            atomicAdd(&vals[slot][2], p2);
          } else {  // a crowded table: integer sums are exact in any order
            atomicAdd(reinterpret_cast<unsigned long long *>(acc_feat + (size_t)ppid * channels + c), p0);
            atomicAdd(reinterpret_cast<unsigned long long *>(acc_pts + (size_t)ppid * 2 + 0), p1);
            atomicAdd(reinterpret_cast<unsigned long long *>(acc_pts + (size_t)ppid * 2 + 1), p2);
          }
        }
        ppid = pid;
        p0 = t0;
        p1 = t1;
        p2 = t2;
      }
      if (valid && background_grad) background_grad[e] = bg;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kAccSlots; i += 256) {
    const int id0 = keys[i];
    if (id0 < 0) continue;
    atomicAdd(reinterpret_cast<unsigned long long *>(acc_feat + (size_t)id0 * channels + c), vals[i][0]);
    atomicAdd(reinterpret_cast<unsigned long long *>(acc_pts + (size_t)id0 * 2 + 0), vals[i][1]);
    atomicAdd(reinterpret_cast<unsigned long long *>(acc_pts + (size_t)id0 * 2 + 1), vals[i][2]);
  }
}

__global__ __launch_bounds__(256) void p2i_max_bwd_finish_kernel(
    const long long *__restrict__ acc_pts, const long long *__restrict__ acc_feat,
    const unsigned *__restrict__ absmax, float *__restrict__ points_grad,
    float *__restrict__ feat_grad, long npts2, long nfeat, float min_radius) {
  const double inv = 1.0 / fixed_scale(absmax, min_radius);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < npts2 + nfeat;
       e += (long)gridDim.x * blockDim.x) {
    if (e < npts2)
      points_grad[e] = (float)((double)acc_pts[e] * inv);
    else
      feat_grad[e - npts2] = (float)((double)acc_feat[e - npts2] * inv);
  }
}

}  // namespace

extern "C" size_t sn_p2i_max_workspace_bytes(int batch, int channels, int h, int w) {
  if (batch < 1 || channels < 1 || h < 1 || w < 1) return 0;
  return (size_t)batch * channels * h * w * 8;
}

namespace {

constexpr float kTileMaxRadius = 16.f;  // larger kernels use the global scatter

size_t tile_workspace_bytes(int npoints, int batch, int h, int w) {
  const size_t cells = (size_t)batch * sn::ceil_div(h, kCell) * sn::ceil_div(w, kCell);
  return 256 + 2 * sn::align_up(cells * 4, 256) + (size_t)npoints * 16;
}

// largest fp32 s with sqrtf(s) <= radius, on the host (IEEE sqrtf is correctly rounded there too)
float max_sq_inside_host(float radius) {
  volatile float s = radius * radius;
  auto next = [](float v, int d) {
    unsigned u;
    memcpy(&u, &v, 4);
    u += d;
    memcpy(&v, &u, 4);
    return v;
  };
  while (sqrtf(s) > radius) s = next(s, -1);
  for (;;) {
    const float t = next(s, 1);
    if (!(sqrtf(t) <= radius)) break;
    s = t;
  }
  return s;
}

// values + ids of nradii splats that share points / features / background (radii <= 16 px)
int tile_forward(const char *fn, const float *points, const float *feat, const int *batch_inds,
                 const float *background, int npoints, int channels, int batch, int h, int w,
                 const float *radii, int nradii, int image_major, float *out, int *out_ids, void *workspace,
                 hipStream_t s) {
  RadiiArg ra = {};
  float rmax = 0.f;
  for (int k = 0; k < nradii; ++k) {
    ra.radius[k] = radii[k];
    ra.s_max[k] = max_sq_inside_host(radii[k]);
    ra.rev_scale[k] = 0.5f / radii[k];  // r*pi/R radians = r/(2R) revolutions
    ra.inv_r2[k] = 1.0f / (radii[k] * radii[k]);
    ra.s_max_all = ra.s_max[k] > ra.s_max_all ? ra.s_max[k] : ra.s_max_all;
    rmax = radii[k] > rmax ? radii[k] : rmax;
  }
  ra.halo = (int)floorf(rmax / kCell) + 1;
  const int cells_x = sn::ceil_div(w, kCell), cells_y = sn::ceil_div(h, kCell);
  const long cells = (long)batch * cells_x * cells_y;
  const long tiles = cells * channels;
  SN_REQUIRE(tiles / 4 + 1 < (1L << 31), "%s: too many tiles", fn);
  char *wp = static_cast<char *>(workspace);
  unsigned *fmax_bits = reinterpret_cast<unsigned *>(wp); wp += 256;   // max |feature| (bits), for the error band
  unsigned *need = fmax_bits + 1;   // != 0: the generic binning kernels have to run
  int *counts = reinterpret_cast<int *>(wp); wp += sn::align_up((size_t)cells * 4, 256);
  int *offs = reinterpret_cast<int *>(wp); wp += sn::align_up((size_t)cells * 4, 256);
  float4 *srec = reinterpret_cast<float4 *>(wp);
  const bool try_grouped = npoints > 0 && npoints % batch == 0 && cells_x * cells_y <= kGroupCells;
  if (try_grouped) {
    SN_HIP(hipMemsetAsync(fmax_bits, 0, 8, s));
    p2i_bin_grouped_kernel<<<batch, 1024, 0, s>>>(points, feat, batch_inds, offs, srec, fmax_bits, need,
                                                  npoints / batch, channels, h, w, cells_x, cells_y);
  } else {
    SN_HIP(hipMemsetAsync(fmax_bits, 0, 4, s));
    SN_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(need), 1, 1, s));
  }
  SN_HIP(hipMemsetAsync(counts, 0, (size_t)cells * 4, s));
  if (npoints > 0)
    p2i_bin_count_kernel<<<lin_blocks(npoints), 256, 0, s>>>(points, batch_inds, counts, npoints, batch,
                                                             h, w, cells_x, cells_y, need);
  p2i_bin_scan_kernel<<<batch, 1024, 0, s>>>(counts, offs, cells_x * cells_y, need);
  if (npoints > 0)
    p2i_bin_scatter_kernel<<<lin_blocks(npoints), 256, 0, s>>>(points, feat, batch_inds, offs, srec, fmax_bits,
                                                               npoints, channels, batch, h, w, cells_x,
                                                               cells_y, need);
  const int blocks = (int)((tiles + 3) / 4);
  const long chw = (long)channels * h * w;
  const long obstride = image_major ? (long)nradii * chw : chw, orstride = image_major ? chw : (long)batch * chw;
#define SN_GATHER(NR)                                                                         \
  do {                                                                                        \
    if (channels == 1)                                                                        \
      p2i_gather_max_kernel<NR, true><<<blocks, 256, 0, s>>>(feat, background, srec, offs,      \
          fmax_bits, channels, batch, h, w, cells_x, cells_y, ra, out, out_ids, obstride, orstride); \
    else                                                                                      \
      p2i_gather_max_kernel<NR, false><<<blocks, 256, 0, s>>>(feat, background, srec, offs,     \
          fmax_bits, channels, batch, h, w, cells_x, cells_y, ra, out, out_ids, obstride, orstride); \
  } while (0)
  if (sn::prof_enabled()) sn::prof_begin("p2i_max_splat", s);
  switch (nradii) {
    case 1: SN_GATHER(1); break;
    case 2: SN_GATHER(2); break;
    case 3: SN_GATHER(3); break;
    default: SN_GATHER(4); break;
  }
  if (sn::prof_enabled()) sn::prof_end("p2i_max_splat", s);
#undef SN_GATHER
  return sn::launch_status(fn);
}

}  // namespace

#ifdef SN_P2I_DIAG
extern "C" int sn_p2i_gather_diag(unsigned long long *out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gather_diag), 64) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gather_diag), z, 64) != hipSuccess) return 1;
  }
  return 0;
}
#endif

// test hook: the two fp32 series of the renderer evaluated on the device (tests pin their error bounds)
extern "C" int sn_p2i_series(const float *u, int n, float *weight, float *slope, void *stream) {
  SN_REQUIRE(u && weight && slope && n >= 0, "sn_p2i_series: bad arguments");
  if (n > 0) p2i_series_kernel<<<(n + 255) / 256, 256, 0, sn::as_stream(stream)>>>(u, n, weight, slope);
  return sn::launch_status("sn_p2i_series");
}

extern "C" size_t sn_p2i_max_multi_workspace_bytes(int npoints, int batch, int channels, int h,
                                                   int w) {
  if (npoints < 0 || batch < 1 || channels < 1 || h < 1 || w < 1) return 0;
  const size_t a = sn_p2i_max_workspace_bytes(batch, channels, h, w);
  const size_t t = tile_workspace_bytes(npoints, batch, h, w);
  return a > t ? a : t;
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
  if (radius <= kTileMaxRadius && workspace_bytes >= tile_workspace_bytes(npoints, batch, h, w))
    return tile_forward("sn_p2i_max_forward", points, feat, batch_inds, background, npoints,
                        channels, batch, h, w, &radius, 1, 0, out, out_ids, workspace, s);
  unsigned long long *img = static_cast<unsigned long long *>(workspace);
  const long px = (long)batch * channels * h * w;
  p2i_max_init_kernel<<<lin_blocks(px), 256, 0, s>>>(background, img, px);
  const long groups = (long)npoints * channels;
  if (groups > 0) {
    const int lpp = lanes_per_point(radius);
    const long blocks = xcd_point_blocks(lpp, npoints, channels, batch);
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

extern "C" int sn_p2i_max_forward_multi(const float *points, const float *feat,
                                        const int *batch_inds, const float *background,
                                        int npoints, int channels, int batch, int h, int w,
                                        const float *radii, int nradii, int image_major, float *out,
                                        int *out_ids, void *workspace, size_t workspace_bytes,
                                        void *stream) {
  SN_REQUIRE(out && out_ids && workspace && radii, "sn_p2i_max_forward_multi: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && batch_inds), "sn_p2i_max_forward_multi: null pointer");
  SN_REQUIRE(nradii >= 1 && nradii <= kMaxRadii, "sn_p2i_max_forward_multi: 1..%d radii (got %d)",
             kMaxRadii, nradii);
  float rmax = 0.f;
  for (int k = 0; k < nradii; ++k) {
    if (int rc = check_common("sn_p2i_max_forward_multi", npoints, channels, batch, h, w, radii[k]))
      return rc;
    rmax = radii[k] > rmax ? radii[k] : rmax;
  }
  SN_REQUIRE(workspace_bytes >= sn_p2i_max_multi_workspace_bytes(npoints, batch, channels, h, w),
             "sn_p2i_max_forward_multi: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  if (rmax <= kTileMaxRadius)
    return tile_forward("sn_p2i_max_forward_multi", points, feat, batch_inds, background, npoints,
                        channels, batch, h, w, radii, nradii, image_major, out, out_ids, workspace, s);
  SN_REQUIRE(!image_major || nradii == 1,
             "sn_p2i_max_forward_multi: the image-major layout needs radii <= %g px", (double)kTileMaxRadius);
  SN_REQUIRE(background, "sn_p2i_max_forward_multi: a null (all-zero) background needs radii <= %g px",
             (double)kTileMaxRadius);
  const size_t image = (size_t)batch * channels * h * w;
  for (int k = 0; k < nradii; ++k)  // large kernels: one global splat per radius
    if (int rc = sn_p2i_max_forward(points, feat, batch_inds, background, npoints, channels, batch,
                                    h, w, radii[k], out + k * image, out_ids + k * image,
                                    workspace, workspace_bytes, stream))
      return rc;
  return 0;
}

extern "C" size_t sn_p2i_max_backward_workspace_bytes(int batch, int channels, int h, int w) {
  if (batch < 1 || channels < 1 || h < 1 || w < 1) return 0;
  return (size_t)batch * channels * h * w * 12;
}

extern "C" int sn_p2i_max_backward(const float *out_grad, const int *out_ids, const float *points,
                                   const float *feat, const int *batch_inds, int npoints,
                                   int channels, int batch, int h, int w, float radius,
                                   float *points_grad, float *feat_grad, float *background_grad,
                                   void *workspace, size_t workspace_bytes, void *stream) {
  SN_REQUIRE(out_grad && out_ids && background_grad && workspace, "sn_p2i_max_backward: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && points_grad && feat_grad),
             "sn_p2i_max_backward: null pointer");
  if (int rc = check_common("sn_p2i_max_backward", npoints, channels, batch, h, w, radius)) return rc;
  SN_REQUIRE(workspace_bytes >= sn_p2i_max_backward_workspace_bytes(batch, channels, h, w),
             "sn_p2i_max_backward: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  const long px = (long)batch * channels * h * w;
  float *contrib = static_cast<float *>(workspace);
  p2i_max_bwd_pixels_kernel<<<lin_blocks(px), 256, 0, s>>>(out_grad, out_ids, points, feat,
                                                           background_grad, contrib, channels, h, w,
                                                           radius, px);
  if (npoints > 0) {
    const int lpp = lanes_per_point_gather(radius);
    const long blocks = xcd_point_blocks(lpp, npoints, 1, batch);
#define SN_BWD(L)                                                                            \
  p2i_max_bwd_points_kernel<L><<<(int)blocks, 256, 0, s>>>(out_ids, contrib, points, batch_inds, \
      points_grad, feat_grad, npoints, channels, batch, h, w, radius, px)
    switch (lpp) {
      case 4: SN_BWD(4); break;
      case 8: SN_BWD(8); break;
      case 16: SN_BWD(16); break;
      case 32: SN_BWD(32); break;
      default: SN_BWD(64); break;
    }
#undef SN_BWD
  }
  return sn::launch_status("sn_p2i_max_backward");
}

extern "C" size_t sn_p2i_max_backward_multi_workspace_bytes(int npoints, int channels) {
  if (npoints < 0 || channels < 1) return 0;
  return 256 + (size_t)npoints * (2 + (size_t)channels) * 8;
}

extern "C" int sn_p2i_max_backward_multi(const float *out_grad, const int *out_ids,
                                         const float *points, const float *feat, int npoints,
                                         int channels, int batch, int h, int w, const float *radii,
                                         int nradii, int image_major, float *points_grad,
                                         float *feat_grad, float *background_grad, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  SN_REQUIRE(out_grad && out_ids && workspace && radii, "sn_p2i_max_backward_multi: null pointer");
  SN_REQUIRE(npoints == 0 || (points && feat && points_grad && feat_grad),
             "sn_p2i_max_backward_multi: null pointer");
  SN_REQUIRE(nradii >= 1 && nradii <= kMaxRadii, "sn_p2i_max_backward_multi: 1..%d radii (got %d)",
             kMaxRadii, nradii);
  RadiiArg ra = {};
  float rmin = 3e38f;
  for (int k = 0; k < nradii; ++k) {
    if (int rc = check_common("sn_p2i_max_backward_multi", npoints, channels, batch, h, w, radii[k]))
      return rc;
    ra.radius[k] = radii[k];
    ra.inv_r2[k] = 1.0f / (radii[k] * radii[k]);
    rmin = radii[k] < rmin ? radii[k] : rmin;
  }
  SN_REQUIRE(workspace_bytes >= sn_p2i_max_backward_multi_workspace_bytes(npoints, channels),
             "sn_p2i_max_backward_multi: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  const long px = (long)batch * channels * h * w;
  unsigned *absmax = static_cast<unsigned *>(workspace);
  long long *acc_pts = reinterpret_cast<long long *>(static_cast<char *>(workspace) + 256);
  long long *acc_feat = acc_pts + (size_t)npoints * 2;
  SN_HIP(hipMemsetAsync(workspace, 0, sn_p2i_max_backward_multi_workspace_bytes(npoints, channels), s));
  p2i_absmax_kernel<<<512, 1024, 0, s>>>(out_grad, px * nradii, feat,
                                                           (long)npoints * channels, absmax);
  // one workgroup per region of kAccRegion x kAccRegion tiles; images b with b % 8 == x on XCD x (blockIdx % 8)
  const long per_image = (long)channels * sn::ceil_div(sn::ceil_div(h, kCell), kAccRegion) *
                         sn::ceil_div(sn::ceil_div(w, kCell), kAccRegion);
  const long blocks = per_image * 8 * ((batch + 7) / 8);
  SN_REQUIRE(blocks < (1L << 31), "sn_p2i_max_backward_multi: image too large");
  p2i_max_bwd_accum_kernel<<<(int)blocks, 256, 0, s>>>(out_grad, out_ids, points, feat, absmax,
                                                       background_grad, acc_pts, acc_feat, channels,
                                                       batch, h, w, ra, nradii, rmin,
                                                       image_major ? (long)nradii * channels * h * w : (long)channels * h * w,
                                                       image_major ? (long)channels * h * w : px);
  if (npoints > 0)
    p2i_max_bwd_finish_kernel<<<lin_blocks((long)npoints * (2 + channels)), 256, 0, s>>>(
        acc_pts, acc_feat, absmax, points_grad, feat_grad, (long)npoints * 2,
        (long)npoints * channels, rmin);
  return sn::launch_status("sn_p2i_max_backward_multi");
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
