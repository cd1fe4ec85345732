// depth_project.hip -- the per-view glue of ComputeDepthMaps as two small kernels each way.
//
// Reference: utils/p2i_utils.py:211-228 (forward of one view), executed there as ~25 torch ops
// per view (expand the 4x4 matrix per point, bmm, divide by w, stack (-y, x), global min / max of
// z, depth feature, NDC -> pixel rescale inside p2i, cuda/p2i_op/__init__.py:117-121) and as many
// autograd nodes on the way back:
//   o      = M [x y z 1]^T,  pos = o.xyz / o.w
//   pixel  = ((-pos.y, pos.x) + 1) / 2 * (S - 1)                (row, col)
//   feat   = 1 - (pos.z - zmin) / (zmax - zmin),  zmin / zmax over the WHOLE tensor
// Backward: the chain rule of exactly these expressions, including the paths through zmin and
// zmax (torch's full-reduction min / max send their gradient evenly to every element that
// attains the extreme).
#include "common.hpp"

namespace {

struct Mat4 {
  float m[16];  // row major
};

__device__ __forceinline__ unsigned ord_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_bits(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// The reference multiplies the 4x4 matrix with [x y z 1]^T as a batched matrix product
// (utils/p2i_utils.py:153-165); its CPU execution -- what the golden vectors were generated with --
// accumulates left to right with separately rounded products and sums, and so does this.
__device__ __forceinline__ void transform(const Mat4 &M, float x, float y, float z, float o[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int r = 0; r < 4; ++r)
    o[r] = ((M.m[r * 4] * x + M.m[r * 4 + 1] * y) + M.m[r * 4 + 2] * z) + M.m[r * 4 + 3];
}

// 1024-thread blocks, at most 128 of them: every block ends with two same-address atomics, which
// serialise at ~25 ns each (2048 blocks of 256 threads spent 45 of their 49 us there)
__global__ __launch_bounds__(1024) void depth_project_kernel(const float *__restrict__ data, long n,
                                                            Mat4 M, float extent,
                                                            float2 *__restrict__ pixel,
                                                            float *__restrict__ zbuf,
                                                            unsigned *__restrict__ zminmax) {
  __shared__ unsigned red[2][16];
  unsigned lo = 0xffffffffu, hi = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float o[4];
    transform(M, data[i * 3 + 0], data[i * 3 + 1], data[i * 3 + 2], o);
    const float px = o[0] / o[3], py = o[1] / o[3], pz = o[2] / o[3];
    pixel[i] = make_float2((-py + 1.f) / 2.f * extent, (px + 1.f) / 2.f * extent);
    zbuf[i] = pz;
    const unsigned k = ord_bits(pz);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
  for (int m = 1; m < 64; m <<= 1) {
    const unsigned a = (unsigned)__shfl_xor((int)lo, m), b = (unsigned)__shfl_xor((int)hi, m);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = lo;
    red[1][threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      lo = red[0][w] < lo ? red[0][w] : lo;
      hi = red[1][w] > hi ? red[1][w] : hi;
    }
    atomicMin(zminmax + 0, lo);
    atomicMax(zminmax + 1, hi);
  }
}

__global__ __launch_bounds__(256) void depth_feature_kernel(const float *__restrict__ zbuf, long n,
                                                            const unsigned *__restrict__ zminmax,
                                                            float *__restrict__ feat) {
  const float zmin = unord_bits(zminmax[0]), zmax = unord_bits(zminmax[1]);
  const float range = zmax - zmin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    feat[i] = 1.0f - (zbuf[i] - zmin) / range;
}

// sums for the zmin / zmax paths: S_a = sum g_f (zmax - z) / r^2, S_c = sum g_f (z - zmin) / r^2,
// and how many points attain each extreme
__global__ __launch_bounds__(256) void depth_reduce_kernel(const float *__restrict__ zbuf,
                                                           const float *__restrict__ g_feat, long n,
                                                           const unsigned *__restrict__ zminmax,
                                                           double *__restrict__ sums,
                                                           unsigned *__restrict__ counts) {
  __shared__ double red[2][4];
  __shared__ unsigned cred[2][4];
  const float zmin = unord_bits(zminmax[0]), zmax = unord_bits(zminmax[1]);
  const double r = (double)zmax - (double)zmin, r2 = r * r;
  double sa = 0.0, sc = 0.0;
  unsigned na = 0, nc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float z = zbuf[i];
    const double g = g_feat[i];
    sa += g * ((double)zmax - (double)z) / r2;
    sc += g * ((double)z - (double)zmin) / r2;
    na += z == zmin;
    nc += z == zmax;
  }
  for (int m = 1; m < 64; m <<= 1) {
    sa += __shfl_xor(sa, m);
    sc += __shfl_xor(sc, m);
    na += (unsigned)__shfl_xor((int)na, m);
    nc += (unsigned)__shfl_xor((int)nc, m);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa;
    red[1][threadIdx.x >> 6] = sc;
    cred[0][threadIdx.x >> 6] = na;
    cred[1][threadIdx.x >> 6] = nc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      sa += red[0][w];
      sc += red[1][w];
      na += cred[0][w];
      nc += cred[1][w];
    }
    atomicAdd(sums + 0, sa);
    atomicAdd(sums + 1, sc);
    if (na) atomicAdd(counts + 0, na);
    if (nc) atomicAdd(counts + 1, nc);
  }
}

__global__ __launch_bounds__(256) void depth_project_bwd_kernel(
    const float *__restrict__ data, long n, Mat4 M, float extent, const float *__restrict__ zbuf,
    const unsigned *__restrict__ zminmax, const float2 *__restrict__ g_pixel,
    const float *__restrict__ g_feat, const double *__restrict__ sums,
    const unsigned *__restrict__ counts, float *__restrict__ g_data) {
  const float zmin = unord_bits(zminmax[0]), zmax = unord_bits(zminmax[1]);
  const float range = zmax - zmin;
  const float ga = g_feat ? (float)(sums[0] / (double)counts[0]) : 0.f;
  const float gc = g_feat ? (float)(sums[1] / (double)counts[1]) : 0.f;
  const float half = extent * 0.5f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float o[4];
    transform(M, data[i * 3 + 0], data[i * 3 + 1], data[i * 3 + 2], o);
    const float z = zbuf[i];
    float gpz = 0.f;
    if (g_feat) gpz = -g_feat[i] / range + (z == zmin ? ga : 0.f) + (z == zmax ? gc : 0.f);
    float gpx = 0.f, gpy = 0.f;
    if (g_pixel) {
      const float2 gp = g_pixel[i];  // d row / d pos.y = -extent/2, d col / d pos.x = extent/2
      gpy = -gp.x * half;
      gpx = gp.y * half;
    }
    const float iw = 1.0f / o[3];
    const float go0 = gpx * iw, go1 = gpy * iw, go2 = gpz * iw;
    const float go3 = -(gpx * o[0] + gpy * o[1] + gpz * o[2]) * iw * iw;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      g_data[i * 3 + c] = M.m[c] * go0 + M.m[4 + c] * go1 + M.m[8 + c] * go2 + M.m[12 + c] * go3;
  }
}

int blocks_for(long n) {
  const long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

__global__ void depth_zminmax_init_kernel(unsigned *zminmax, int nviews) {
  const int i = threadIdx.x;
  if (i < 2 * nviews) zminmax[i] = (i & 1) ? 0u : 0xffffffffu;
}

extern "C" int sn_depth_project_forward(const float *data, long npoints, const float *matrix16,
                                        float extent, float *pixel, float *z,
                                        unsigned *zminmax, float *feat, void *stream) {
  SN_REQUIRE(matrix16 && zminmax, "sn_depth_project_forward: null pointer");
  SN_REQUIRE(npoints >= 0, "sn_depth_project_forward: npoints < 0");
  hipStream_t s = sn::as_stream(stream);
  // {min key, max key} = {0xffffffff, 0}: two memset nodes, no pageable host copy (capturable, never blocks)
  SN_HIP(hipMemsetAsync(zminmax, 0xff, 4, s));
  SN_HIP(hipMemsetAsync(zminmax + 1, 0, 4, s));
  if (npoints == 0) return 0;
  SN_REQUIRE(data && pixel && z && feat, "sn_depth_project_forward: null pointer");
  Mat4 M;
  for (int i = 0; i < 16; ++i) M.m[i] = matrix16[i];
  const long pb = (npoints + 1023) / 1024;
  depth_project_kernel<<<(int)(pb > 128 ? 128 : pb), 1024, 0, s>>>(data, npoints, M, extent,
                                                         reinterpret_cast<float2 *>(pixel), z, zminmax);
  depth_feature_kernel<<<blocks_for(npoints), 256, 0, s>>>(z, npoints, zminmax, feat);
  return sn::launch_status("sn_depth_project_forward");
}

extern "C" int sn_depth_project_backward(const float *data, long npoints, const float *matrix16,
                                         float extent, const float *z, const unsigned *zminmax,
                                         const float *g_pixel, const float *g_feat,
                                         void *workspace32, float *g_data, void *stream) {
  SN_REQUIRE(matrix16 && zminmax && workspace32, "sn_depth_project_backward: null pointer");
  if (npoints == 0) return 0;
  SN_REQUIRE(data && z && g_data, "sn_depth_project_backward: null pointer");
  hipStream_t s = sn::as_stream(stream);
  Mat4 M;
  for (int i = 0; i < 16; ++i) M.m[i] = matrix16[i];
  double *sums = static_cast<double *>(workspace32);
  unsigned *counts = reinterpret_cast<unsigned *>(sums + 2);
  if (g_feat) {
    SN_HIP(hipMemsetAsync(workspace32, 0, 32, s));
    depth_reduce_kernel<<<blocks_for(npoints) < 96 ? blocks_for(npoints) : 96, 256, 0, s>>>(
        z, g_feat, npoints, zminmax, sums, counts);
  }
  depth_project_bwd_kernel<<<blocks_for(npoints), 256, 0, s>>>(
      data, npoints, M, extent, z, zminmax, reinterpret_cast<const float2 *>(g_pixel), g_feat, sums,
      counts, g_data);
  return sn::launch_status("sn_depth_project_backward");
}


// ---------------------------------------------------------------------------------------------------
// All views of a ComputeDepthMaps sweep in one set of launches.  The per-view path above costs ~10 small
// launches per view (projection, feature, binning, reductions ...), each far too short to fill the chip;
// a renderer call over V views is V x B independent images, so the views simply become part of the batch:
// pixel / z / feat are [V, n], zminmax [V, 2] (the depth normalisation stays PER VIEW over the whole input
// tensor, utils/p2i_utils.py:226), and the splat sees V x B images.  blockIdx.y = view.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kMaxViews = 8;
struct MatV {
  Mat4 v[kMaxViews];
};

__global__ __launch_bounds__(1024) void depth_project_views_kernel(const float *__restrict__ data, long n, MatV M,
                                                                  float extent, float2 *__restrict__ pixel,
                                                                  float *__restrict__ zbuf,
                                                                  unsigned *__restrict__ zminmax) {
  __shared__ unsigned red[2][16];
  const int view = blockIdx.y;
  const Mat4 &Mv = M.v[view];
  pixel += (size_t)view * n;
  zbuf += (size_t)view * n;
  unsigned lo = 0xffffffffu, hi = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float o[4];
    transform(Mv, data[i * 3 + 0], data[i * 3 + 1], data[i * 3 + 2], o);
    const float px = o[0] / o[3], py = o[1] / o[3], pz = o[2] / o[3];
    pixel[i] = make_float2((-py + 1.f) / 2.f * extent, (px + 1.f) / 2.f * extent);
    zbuf[i] = pz;
    const unsigned k = ord_bits(pz);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
  for (int m = 1; m < 64; m <<= 1) {
    const unsigned a = (unsigned)__shfl_xor((int)lo, m), b = (unsigned)__shfl_xor((int)hi, m);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = lo;
    red[1][threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      lo = red[0][w] < lo ? red[0][w] : lo;
      hi = red[1][w] > hi ? red[1][w] : hi;
    }
    atomicMin(zminmax + 2 * view + 0, lo);
    atomicMax(zminmax + 2 * view + 1, hi);
  }
}

__global__ __launch_bounds__(256) void depth_feature_views_kernel(const float *__restrict__ zbuf, long n,
                                                                  const unsigned *__restrict__ zminmax,
                                                                  float *__restrict__ feat) {
  const int view = blockIdx.y;
  const float zmin = unord_bits(zminmax[2 * view]), zmax = unord_bits(zminmax[2 * view + 1]);
  const float range = zmax - zmin;
  zbuf += (size_t)view * n;
  feat += (size_t)view * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    feat[i] = 1.0f - (zbuf[i] - zmin) / range;
}

__global__ __launch_bounds__(256) void depth_reduce_views_kernel(const float *__restrict__ zbuf,
                                                                 const float *__restrict__ g_feat, long n,
                                                                 const unsigned *__restrict__ zminmax,
                                                                 double *__restrict__ sums_all) {
  __shared__ double red[2][4];
  __shared__ unsigned cred[2][4];
  const int view = blockIdx.y;
  double *sums = sums_all + 4 * view;                      // {S_a, S_c, counts as two unsigned in one double slot...}
  unsigned *counts = reinterpret_cast<unsigned *>(sums + 2);
  zbuf += (size_t)view * n;
  g_feat += (size_t)view * n;
  const float zmin = unord_bits(zminmax[2 * view]), zmax = unord_bits(zminmax[2 * view + 1]);
  const double r = (double)zmax - (double)zmin, r2 = r * r;
  double sa = 0.0, sc = 0.0;
  unsigned na = 0, nc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float z = zbuf[i];
    const double g = g_feat[i];
    sa += g * ((double)zmax - (double)z) / r2;
    sc += g * ((double)z - (double)zmin) / r2;
    na += z == zmin;
    nc += z == zmax;
  }
  for (int m = 1; m < 64; m <<= 1) {
    sa += __shfl_xor(sa, m);
    sc += __shfl_xor(sc, m);
    na += (unsigned)__shfl_xor((int)na, m);
    nc += (unsigned)__shfl_xor((int)nc, m);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa;
    red[1][threadIdx.x >> 6] = sc;
    cred[0][threadIdx.x >> 6] = na;
    cred[1][threadIdx.x >> 6] = nc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      sa += red[0][w];
      sc += red[1][w];
      na += cred[0][w];
      nc += cred[1][w];
    }
    atomicAdd(sums + 0, sa);
    atomicAdd(sums + 1, sc);
    if (na) atomicAdd(counts + 0, na);
    if (nc) atomicAdd(counts + 1, nc);
  }
}

// one thread per point: the contributions of all views meet in registers, one store
__global__ __launch_bounds__(256) void depth_project_bwd_views_kernel(
    const float *__restrict__ data, long n, MatV M, int nviews, float extent, const float *__restrict__ zbuf,
    const unsigned *__restrict__ zminmax, const float2 *__restrict__ g_pixel, const float *__restrict__ g_feat,
    const double *__restrict__ sums_all, float *__restrict__ g_data) {
  const float half = extent * 0.5f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = data[i * 3 + 0], y = data[i * 3 + 1], zc = data[i * 3 + 2];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int v = 0; v < nviews; ++v) {
      const Mat4 &Mv = M.v[v];
      const float zmin = unord_bits(zminmax[2 * v]), zmax = unord_bits(zminmax[2 * v + 1]);
      const float range = zmax - zmin;
      const double *sums = sums_all + 4 * v;
      const unsigned *counts = reinterpret_cast<const unsigned *>(sums + 2);
      float o[4];
      transform(Mv, x, y, zc, o);
      const float z = zbuf[(size_t)v * n + i];
      float gpz = 0.f;
      if (g_feat) {
        const float ga = (float)(sums[0] / (double)counts[0]), gc = (float)(sums[1] / (double)counts[1]);
        gpz = -g_feat[(size_t)v * n + i] / range + (z == zmin ? ga : 0.f) + (z == zmax ? gc : 0.f);
      }
      float gpx = 0.f, gpy = 0.f;
      if (g_pixel) {
        const float2 gp = g_pixel[(size_t)v * n + i];
        gpy = -gp.x * half;
        gpx = gp.y * half;
      }
      const float iw = 1.0f / o[3];
      const float go0 = gpx * iw, go1 = gpy * iw, go2 = gpz * iw;
      const float go3 = -(gpx * o[0] + gpy * o[1] + gpz * o[2]) * iw * iw;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        acc[c] += Mv.m[c] * go0 + Mv.m[4 + c] * go1 + Mv.m[8 + c] * go2 + Mv.m[12 + c] * go3;
    }
    g_data[i * 3 + 0] = acc[0];
    g_data[i * 3 + 1] = acc[1];
    g_data[i * 3 + 2] = acc[2];
  }
}

}  // namespace

extern "C" int sn_depth_project_forward_views(const float *data, long npoints, const float *matrices16,
                                              int nviews, float extent, float *pixel, float *z,
                                              unsigned *zminmax, float *feat, void *stream) {
  SN_REQUIRE(matrices16 && zminmax, "sn_depth_project_forward_views: null pointer");
  SN_REQUIRE(npoints >= 0 && nviews >= 1 && nviews <= kMaxViews,
             "sn_depth_project_forward_views: need 1 <= nviews <= 8 (got %d)", nviews);
  hipStream_t s = sn::as_stream(stream);
  // {min key, max key} = {0xffffffff, 0} per view: ONE tiny launch (two 4-byte memsets per view were 16 dependent
  // 5 us nodes in front of every sweep)
  depth_zminmax_init_kernel<<<1, 64, 0, s>>>(zminmax, nviews);
  if (npoints == 0) return 0;
  SN_REQUIRE(data && pixel && z && feat, "sn_depth_project_forward_views: null pointer");
  MatV M;
  for (int v = 0; v < nviews; ++v)
    for (int i = 0; i < 16; ++i) M.v[v].m[i] = matrices16[v * 16 + i];
  const long pb = (npoints + 1023) / 1024;
  depth_project_views_kernel<<<dim3((unsigned)(pb > 64 ? 64 : pb), nviews), 1024, 0, s>>>(
      data, npoints, M, extent, reinterpret_cast<float2 *>(pixel), z, zminmax);
  depth_feature_views_kernel<<<dim3(blocks_for(npoints), nviews), 256, 0, s>>>(z, npoints, zminmax, feat);
  return sn::launch_status("sn_depth_project_forward_views");
}

extern "C" int sn_depth_project_backward_views(const float *data, long npoints, const float *matrices16,
                                               int nviews, float extent, const float *z,
                                               const unsigned *zminmax, const float *g_pixel,
                                               const float *g_feat, void *workspace256, float *g_data,
                                               void *stream) {
  SN_REQUIRE(matrices16 && zminmax && workspace256, "sn_depth_project_backward_views: null pointer");
  SN_REQUIRE(nviews >= 1 && nviews <= kMaxViews, "sn_depth_project_backward_views: need 1 <= nviews <= 8");
  if (npoints == 0) return 0;
  SN_REQUIRE(data && z && g_data, "sn_depth_project_backward_views: null pointer");
  hipStream_t s = sn::as_stream(stream);
  MatV M;
  for (int v = 0; v < nviews; ++v)
    for (int i = 0; i < 16; ++i) M.v[v].m[i] = matrices16[v * 16 + i];
  double *sums = static_cast<double *>(workspace256);   // per view: {S_a, S_c, (n_a, n_c), pad}
  if (g_feat) {
    SN_HIP(hipMemsetAsync(workspace256, 0, 32 * (size_t)nviews, s));
    const int rb = blocks_for(npoints) < 48 ? blocks_for(npoints) : 48;
    depth_reduce_views_kernel<<<dim3(rb, nviews), 256, 0, s>>>(z, g_feat, npoints, zminmax, sums);
  }
  depth_project_bwd_views_kernel<<<blocks_for(npoints), 256, 0, s>>>(
      data, npoints, M, nviews, extent, z, zminmax, reinterpret_cast<const float2 *>(g_pixel), g_feat, sums,
      g_data);
  return sn::launch_status("sn_depth_project_backward_views");
}
