// gridding.hip -- GRNet gridding / gridding-reverse / cubic feature sampling (gfx950).
//
// Reference: cuda/gridding/gridding.cu:29-177 (fwd), :213-312 (bwd);
// cuda/gridding/gridding_reverse.cu:30-103 (fwd), :124-214 (bwd);
// cuda/cubic_feature_sampling/cubic_feature_sampling.cu:29-102 (fwd), :135-174 (bwd).
// Semantics: oracle/gridding.c.
//
// MI355X design: the reference launches ONE block of <= 512 threads per sample
// (B blocks on a 256-CU chip); these are bandwidth / atomic bound scatter-gather ops,
// so here every kernel is a flat grid-stride launch over (sample x point) or
// (sample x vertex): thousands of workgroups, coalesced point reads, fp32 hardware
// atomics into the 1 MiB-per-sample grid that stays in L2.  The cubic gather puts the
// channel index on adjacent lanes so the [.., C] output rows are written coalesced.
#include "common.hpp"

namespace {

__device__ __forceinline__ void corners(float p, int &lo, int &up) {
  lo = (int)floorf(p);
  up = (int)ceilf(p);
  if (lo == up) up += 1;
}

int lin_blocks(long total, int threads = 256) {
  const long b = (total + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

__global__ __launch_bounds__(256) void gridding_fwd_kernel(int npts, int s, int nverts,
                                                           const float *__restrict__ ptcloud,
                                                           float *__restrict__ grid,
                                                           float *__restrict__ weights,
                                                           int *__restrict__ indexes, long total,
                                                           int skip_zero_rows) {
  const int len = 2 * s;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / npts;
    const float px = ptcloud[e * 3 + 0], py = ptcloud[e * 3 + 1], pz = ptcloud[e * 3 + 2];
    if (skip_zero_rows && (px + py) + pz == 0.f) {  // padding row (cuda/gridding/__init__.py:43-46)
      for (int c = 0; c < 24; ++c) weights[e * 24 + c] = 0.f;
      for (int c = 0; c < 8; ++c) indexes[e * 8 + c] = -1;
      continue;
    }
    int lx, ux, ly, uy, lz, uz;
    corners(px, lx, ux);
    corners(py, ly, uy);
    corners(pz, lz, uz);
    float *w = weights + e * 24;
    int *ix = indexes + e * 8;
    float *g = grid + b * nverts;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int cx = (c & 4) ? ux : lx, cy = (c & 2) ? uy : ly, cz = (c & 1) ? uz : lz;
      const int idx = (cx + s) * len * len + (cy + s) * len + (cz + s);
      const float wx = 1 - fabsf(px - cx), wy = 1 - fabsf(py - cy), wz = 1 - fabsf(pz - cz);
      ix[c] = idx;
      w[c * 3 + 0] = wx;
      w[c * 3 + 1] = wy;
      w[c * 3 + 2] = wz;
      if (idx >= 0 && idx < nverts) unsafeAtomicAdd(g + idx, wx * wy * wz);
    }
  }
}

// gridding distance (cuda/gridding_loss/gridding_distance.cu:29-177): the same trilinear
// weights over an arbitrary integer box [min, max] per axis, but every vertex keeps EIGHT
// accumulators, one per corner role: slot = vertex * 8 + c.  The backward is
// gridding_bwd_kernel with nverts * 8 slots (the reference's grad kernel :214-314 reads the
// slots the forward recorded).
__global__ __launch_bounds__(256) void gridding_dist_fwd_kernel(
    int npts, int min_x, int min_y, int min_z, int len_y, int len_z, int nslots,
    const float *__restrict__ ptcloud, float *__restrict__ grid, float *__restrict__ weights,
    int *__restrict__ indexes, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / npts;
    const float px = ptcloud[e * 3 + 0], py = ptcloud[e * 3 + 1], pz = ptcloud[e * 3 + 2];
    int lx, ux, ly, uy, lz, uz;
    corners(px, lx, ux);
    corners(py, ly, uy);
    corners(pz, lz, uz);
    float *w = weights + e * 24;
    int *ix = indexes + e * 8;
    float *g = grid + b * nslots;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int cx = (c & 4) ? ux : lx, cy = (c & 2) ? uy : ly, cz = (c & 1) ? uz : lz;
      const int idx = (((cx - min_x) * len_y + (cy - min_y)) * len_z + (cz - min_z)) * 8 + c;
      const float wx = 1 - fabsf(px - cx), wy = 1 - fabsf(py - cy), wz = 1 - fabsf(pz - cz);
      ix[c] = idx;
      w[c * 3 + 0] = wx;
      w[c * 3 + 1] = wy;
      w[c * 3 + 2] = wz;
      if (idx >= 0 && idx < nslots) unsafeAtomicAdd(g + idx, wx * wy * wz);
    }
  }
}

__global__ __launch_bounds__(256) void gridding_bwd_kernel(int npts, int nverts,
                                                           const float *__restrict__ grad_grid,
                                                           const float *__restrict__ weights,
                                                           const int *__restrict__ indexes,
                                                           float *__restrict__ grad_ptcloud,
                                                           long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / npts;
    const float *w = weights + e * 24;
    const int *ix = indexes + e * 8;
    const float *gg = grad_grid + b * nverts;
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int idx = ix[c];
      const float g = (idx >= 0 && idx < nverts) ? gg[idx] : 0.f;
      const float wx = w[c * 3], wy = w[c * 3 + 1], wz = w[c * 3 + 2];
      const float sx = (c & 4) ? g : -g, sy = (c & 2) ? g : -g, sz = (c & 1) ? g : -g;
      gx += sx * wy * wz;
      gy += sy * wx * wz;
      gz += sz * wx * wy;
    }
    grad_ptcloud[e * 3 + 0] = gx;
    grad_ptcloud[e * 3 + 1] = gy;
    grad_ptcloud[e * 3 + 2] = gz;
  }
}

struct RevCell {
  int idx[8];
  float w[8];
  float wsum;
  int off[3];
};

__device__ __forceinline__ bool rev_setup(const float *__restrict__ g, int j, int scale, RevCell &c) {
  const int sq = scale * scale;
  const int x = j / sq, y = j % sq / scale, z = j % sq % scale;
  if (x == 0 || y == 0 || z == 0) return false;
  const int base = (x - 1) * sq + (y - 1) * scale + (z - 1);
  c.idx[0] = base;
  c.idx[1] = base + 1;
  c.idx[2] = base + scale;
  c.idx[3] = base + scale + 1;
  c.idx[4] = base + sq;
  c.idx[5] = base + sq + 1;
  c.idx[6] = base + sq + scale;
  c.idx[7] = j;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c.w[i] = g[c.idx[i]];
    s += c.w[i];
  }
  if ((double)s < 1e-6) return false;
#pragma unroll
  for (int i = 0; i < 8; ++i) c.w[i] /= s;
  c.wsum = s;
  c.off[0] = x - scale / 2;
  c.off[1] = y - scale / 2;
  c.off[2] = z - scale / 2;
  return true;
}

__global__ __launch_bounds__(256) void gridding_rev_fwd_kernel(int scale, int n3,
                                                               const float *__restrict__ grid,
                                                               float *__restrict__ ptcloud,
                                                               long total) {
#pragma clang fp contract(off)
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / n3;
    const int j = (int)(e - b * n3);
    float o[3] = {0.f, 0.f, 0.f};
    RevCell c;
    if (rev_setup(grid + b * n3, j, scale, c)) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int hi = a == 0 ? (k & 4) : (a == 1 ? (k & 2) : (k & 1));
          const float coord = (float)(hi ? c.off[a] : c.off[a] - 1);
          acc = k == 0 ? c.w[k] * coord : acc + c.w[k] * coord;
        }
        o[a] = acc;
      }
    }
    ptcloud[e * 3 + 0] = o[0];
    ptcloud[e * 3 + 1] = o[1];
    ptcloud[e * 3 + 2] = o[2];
  }
}

__global__ __launch_bounds__(256) void gridding_rev_bwd_kernel(
    int scale, int n3, const float *__restrict__ grad_ptcloud, const float *__restrict__ grid,
    const float *__restrict__ ptcloud, float *__restrict__ grad_grid, long total) {
#pragma clang fp contract(off)
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / n3;
    const int j = (int)(e - b * n3);
    RevCell c;
    if (!rev_setup(grid + b * n3, j, scale, c)) continue;
    const float g0 = grad_ptcloud[e * 3], g1 = grad_ptcloud[e * 3 + 1], g2 = grad_ptcloud[e * 3 + 2];
    const float p0 = ptcloud[e * 3], p1 = ptcloud[e * 3 + 1], p2 = ptcloud[e * 3 + 2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float cx = (float)((k & 4) ? c.off[0] : c.off[0] - 1) - p0;
      const float cy = (float)((k & 2) ? c.off[1] : c.off[1] - 1) - p1;
      const float cz = (float)((k & 1) ? c.off[2] : c.off[2] - 1) - p2;
      unsafeAtomicAdd(grad_grid + b * n3 + c.idx[k],
                      (g0 * cx / c.wsum + g1 * cy / c.wsum) + g2 * cz / c.wsum);
    }
  }
}

// indexes: one thread per (sample, point)
__global__ __launch_bounds__(256) void cubic_index_kernel(int npts, int scale, int ns, int nv,
                                                          const float *__restrict__ ptcloud,
                                                          int *__restrict__ idx, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    int lo[3], up[3];
    corners(ptcloud[e * 3 + 0], lo[0], up[0]);
    corners(ptcloud[e * 3 + 1], lo[1], up[1]);
    corners(ptcloud[e * 3 + 2], lo[2], up[2]);
    int *ix = idx + e * nv;
    const int ext = ns - 1;
    int v = 0;
    for (int j = lo[0] - ext; j <= up[0] + ext; ++j)
      for (int k = lo[1] - ext; k <= up[1] + ext; ++k)
        for (int m = lo[2] - ext; m <= up[2] + ext; ++m)
          ix[v++] = (j < 0 || j >= scale || k < 0 || k >= scale || m < 0 || m >= scale)
                        ? -1
                        : (j * scale + k) * scale + m;
  }
}

// gather: one thread per output element [b, p, v, k] (k fastest => coalesced stores)
__global__ __launch_bounds__(256) void cubic_gather_kernel(int npts, int c, int cub, int nv,
                                                           const float *__restrict__ feat,
                                                           const int *__restrict__ idx,
                                                           float *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % c);
    const long pv = e / c;               // (b * npts + p) * nv + v
    const long b = pv / ((long)npts * nv);
    const int vtx = idx[pv];
    out[e] = vtx == -1 ? 0.f : feat[(b * c + k) * cub + vtx];
  }
}

__global__ __launch_bounds__(256) void cubic_scatter_kernel(int npts, int c, int cub, int nv,
                                                            const float *__restrict__ grad_out,
                                                            const int *__restrict__ idx,
                                                            float *__restrict__ grad_feat,
                                                            long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % c);
    const long pv = e / c;
    const long b = pv / ((long)npts * nv);
    const int vtx = idx[pv];
    if (vtx != -1) unsafeAtomicAdd(grad_feat + (b * c + k) * cub + vtx, grad_out[e]);
  }
}

}  // namespace

extern "C" int sn_gridding_forward(const float *ptcloud, int b, int npts, int scale, float *grid,
                                   float *weights, int *indexes, void *stream) {
  SN_REQUIRE(grid, "sn_gridding_forward: null pointer");
  SN_REQUIRE(b >= 1 && npts >= 0 && scale >= 2, "sn_gridding_forward: bad sizes");
  SN_REQUIRE(npts == 0 || (ptcloud && weights && indexes), "sn_gridding_forward: null pointer");
  const int s = scale / 2, nverts = 8 * s * s * s;
  hipStream_t st = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(grid, 0, (size_t)b * nverts * 4, st));
  const long total = (long)b * npts;
  if (total > 0)
    gridding_fwd_kernel<<<lin_blocks(total), 256, 0, st>>>(npts, s, nverts, ptcloud, grid, weights,
                                                           indexes, total, 0);
  return sn::launch_status("sn_gridding_forward");
}

extern "C" int sn_gridding_forward_padded(const float *ptcloud, int b, int npts, int scale,
                                          float *grid, float *weights, int *indexes, void *stream) {
  SN_REQUIRE(grid, "sn_gridding_forward_padded: null pointer");
  SN_REQUIRE(b >= 1 && npts >= 0 && scale >= 2 && scale % 2 == 0 && scale <= 1024,
             "sn_gridding_forward_padded: bad sizes");
  const int s = scale / 2, nverts = 8 * s * s * s;
  hipStream_t st = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(grid, 0, (size_t)b * nverts * 4, st));
  const long total = (long)b * npts;
  if (total > 0) {
    SN_REQUIRE(ptcloud && weights && indexes, "sn_gridding_forward_padded: null pointer");
    gridding_fwd_kernel<<<lin_blocks(total), 256, 0, st>>>(npts, s, nverts, ptcloud, grid, weights,
                                                           indexes, total, 1);
  }
  return sn::launch_status("sn_gridding_forward_padded");
}

extern "C" int sn_gridding_dist_forward(const float *ptcloud, int b, int npts, int min_x, int max_x,
                                        int min_y, int max_y, int min_z, int max_z, float *grid,
                                        float *weights, int *indexes, void *stream) {
  SN_REQUIRE(grid, "sn_gridding_dist_forward: null pointer");
  SN_REQUIRE(b >= 1 && npts >= 0 && max_x >= min_x && max_y >= min_y && max_z >= min_z,
             "sn_gridding_dist_forward: bad sizes / bounds");
  const long lx = (long)max_x - min_x + 1, ly = (long)max_y - min_y + 1, lz = (long)max_z - min_z + 1;
  SN_REQUIRE(lx * ly * lz * 8 < (1L << 31), "sn_gridding_dist_forward: grid too large");
  const int nslots = (int)(lx * ly * lz * 8);
  hipStream_t st = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(grid, 0, (size_t)b * nslots * 4, st));
  const long total = (long)b * npts;
  if (total == 0) return 0;
  SN_REQUIRE(ptcloud && weights && indexes, "sn_gridding_dist_forward: null pointer");
  gridding_dist_fwd_kernel<<<lin_blocks(total), 256, 0, st>>>(npts, min_x, min_y, min_z, (int)ly, (int)lz,
                                                              nslots, ptcloud, grid, weights, indexes,
                                                              total);
  return sn::launch_status("sn_gridding_dist_forward");
}

extern "C" int sn_gridding_backward(const float *grad_grid, const float *weights,
                                    const int *indexes, int b, int npts, int nverts,
                                    float *grad_ptcloud, void *stream) {
  SN_REQUIRE(b >= 1 && npts >= 0 && nverts >= 1, "sn_gridding_backward: bad sizes");
  const long total = (long)b * npts;
  if (total == 0) return 0;
  SN_REQUIRE(grad_grid && weights && indexes && grad_ptcloud, "sn_gridding_backward: null pointer");
  gridding_bwd_kernel<<<lin_blocks(total), 256, 0, sn::as_stream(stream)>>>(
      npts, nverts, grad_grid, weights, indexes, grad_ptcloud, total);
  return sn::launch_status("sn_gridding_backward");
}

extern "C" int sn_gridding_reverse_forward(const float *grid, int b, int scale, float *ptcloud,
                                           void *stream) {
  SN_REQUIRE(grid && ptcloud, "sn_gridding_reverse_forward: null pointer");
  SN_REQUIRE(b >= 1 && scale >= 1 && scale <= 1024, "sn_gridding_reverse_forward: bad sizes");
  const int n3 = scale * scale * scale;
  const long total = (long)b * n3;
  gridding_rev_fwd_kernel<<<lin_blocks(total), 256, 0, sn::as_stream(stream)>>>(scale, n3, grid,
                                                                                ptcloud, total);
  return sn::launch_status("sn_gridding_reverse_forward");
}

extern "C" int sn_gridding_reverse_backward(const float *grad_ptcloud, const float *grid,
                                            const float *ptcloud, int b, int scale,
                                            float *grad_grid, void *stream) {
  SN_REQUIRE(grad_ptcloud && grid && ptcloud && grad_grid, "sn_gridding_reverse_backward: null pointer");
  SN_REQUIRE(b >= 1 && scale >= 1 && scale <= 1024, "sn_gridding_reverse_backward: bad sizes");
  const int n3 = scale * scale * scale;
  const long total = (long)b * n3;
  hipStream_t st = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(grad_grid, 0, (size_t)total * 4, st));
  gridding_rev_bwd_kernel<<<lin_blocks(total), 256, 0, st>>>(scale, n3, grad_ptcloud, grid, ptcloud,
                                                             grad_grid, total);
  return sn::launch_status("sn_gridding_reverse_backward");
}

extern "C" int sn_cubic_forward(const float *ptcloud, const float *feat, int b, int npts, int c,
                                int scale, int ns, float *out, int *idx, void *stream) {
  SN_REQUIRE(b >= 1 && npts >= 0 && c >= 1 && scale >= 1 && ns >= 1 && ns <= 8,
             "sn_cubic_forward: bad sizes");
  const long pts = (long)b * npts;
  if (pts == 0) return 0;
  SN_REQUIRE(ptcloud && feat && out && idx, "sn_cubic_forward: null pointer");
  const int nv = 8 * ns * ns * ns, cub = scale * scale * scale;
  hipStream_t st = sn::as_stream(stream);
  cubic_index_kernel<<<lin_blocks(pts), 256, 0, st>>>(npts, scale, ns, nv, ptcloud, idx, pts);
  const long total = pts * nv * c;
  cubic_gather_kernel<<<lin_blocks(total), 256, 0, st>>>(npts, c, cub, nv, feat, idx, out, total);
  return sn::launch_status("sn_cubic_forward");
}

extern "C" int sn_cubic_backward(const float *grad_out, const int *idx, int b, int npts, int c,
                                 int scale, int ns, float *grad_feat, void *stream) {
  SN_REQUIRE(grad_feat, "sn_cubic_backward: null pointer");
  SN_REQUIRE(b >= 1 && npts >= 0 && c >= 1 && scale >= 1 && ns >= 1 && ns <= 8,
             "sn_cubic_backward: bad sizes");
  const int nv = 8 * ns * ns * ns, cub = scale * scale * scale;
  hipStream_t st = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(grad_feat, 0, (size_t)b * c * cub * 4, st));
  const long total = (long)b * npts * nv * c;
  if (total == 0) return 0;
  SN_REQUIRE(grad_out && idx, "sn_cubic_backward: null pointer");
  cubic_scatter_kernel<<<lin_blocks(total), 256, 0, st>>>(npts, c, cub, nv, grad_out, idx, grad_feat,
                                                          total);
  return sn::launch_status("sn_cubic_backward");
}
