// cloud_sort.hpp -- counting sort of a point cloud into the 16^3 cells of a Hilbert curve (gfx950 only).
// Shared by MDS (cluster-sorted slots) and EMD (spatially ordered target stream, seeds).
// Two launches per batch of clouds:
//   cloud_sort_count_kernel    one workgroup per cloud: bounding box, cell of every point,
//                              exclusive scan of the 4096 cell counts -> hist = cell START offsets
//   cloud_sort_scatter_kernel  perm[sorted position] = original index; afterwards hist holds the
//                              cell END offsets (the order inside a cell is arbitrary)
#pragma once
#include "common.hpp"

namespace {

constexpr int kSortCells = 4096;  // 16^3 cells

// Cell code along a 3-D Hilbert curve over the 16^3 grid (Skilling's axes-to-transpose form, then the
// bits interleaved): consecutive codes are face-adjacent cells, so a run of sorted points -- an MDS slot,
// an EMD / Chamfer block of 16 or 64 -- is one compact blob.  Z order jumps at every octant boundary,
// which stretched the bounding boxes the pruning tests use.  (The name is kept: every caller only needs
// "the sort key of a cell".)
__device__ __forceinline__ unsigned morton3_4bit(unsigned x, unsigned y, unsigned z) {
#ifdef SN_SORT_Z_ORDER
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r |= (((x >> i) & 1u) << (3 * i)) | (((y >> i) & 1u) << (3 * i + 1)) | (((z >> i) & 1u) << (3 * i + 2));
  return r;
#else
  unsigned X[3] = {x, y, z};
#pragma unroll
  for (unsigned q = 8u; q > 1u; q >>= 1) {
    const unsigned p = q - 1u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & q) {
        X[0] ^= p;
      } else {
        const unsigned t = (X[0] ^ X[i]) & p;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned t = 0;
#pragma unroll
  for (unsigned q = 8u; q > 1u; q >>= 1)
    if (X[2] & q) t ^= q - 1u;
  X[0] ^= t;
  X[1] ^= t;
  X[2] ^= t;
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r |= (((X[0] >> i) & 1u) << (3 * i + 2)) | (((X[1] >> i) & 1u) << (3 * i + 1)) | (((X[2] >> i) & 1u) << (3 * i));
  return r;
#endif
}

// per cloud: bounding box -> cell histogram (one workgroup per cloud)
__global__ __launch_bounds__(1024) void cloud_sort_count_kernel(int n, const float *__restrict__ xyz,
                                                              float *__restrict__ bbox,
                                                              int *__restrict__ hist,
                                                              int *__restrict__ cell_of) {
  __shared__ float red[6][16];
  __shared__ int lh[kSortCells];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *p = xyz + (size_t)b * n * 3;
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
  for (int k = tid; k < n; k += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[k * 3 + a];
      lo[a] = __builtin_fminf(lo[a], v);
      hi[a] = __builtin_fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int m = 1; m < 64; m <<= 1) {
      lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], m));
      hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], m));
    }
  if ((tid & 63) == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[a][tid >> 6] = lo[a];
      red[3 + a][tid >> 6] = hi[a];
    }
  for (int c = tid; c < kSortCells; c += 1024) lh[c] = 0;
  __syncthreads();
  float blo[3], scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = red[a][0], h = red[3 + a][0];
    for (int w = 1; w < 16; ++w) {
      l = __builtin_fminf(l, red[a][w]);
      h = __builtin_fmaxf(h, red[3 + a][w]);
    }
    blo[a] = l;
    scale[a] = h > l ? 15.999f / (h - l) : 0.f;
    if (tid == 0) {
      bbox[b * 6 + a] = l;
      bbox[b * 6 + 3 + a] = h;
    }
  }
  for (int k = tid; k < n; k += 1024) {
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float f = (p[k * 3 + a] - blo[a]) * scale[a];
      q[a] = (unsigned)(f < 0.f ? 0.f : (f > 15.f ? 15.f : f));
    }
    const int c = (int)morton3_4bit(q[0], q[1], q[2]);
    cell_of[(size_t)b * n + k] = c;
    atomicAdd(&lh[c], 1);
  }
  __syncthreads();
  // exclusive scan of the 4096 cell counts (4 per lane) -> start offsets
  const int c0 = tid * 4;
  const int v0 = lh[c0], v1 = lh[c0 + 1], v2 = lh[c0 + 2], v3 = lh[c0 + 3];
  int sum = v0 + v1 + v2 + v3, incl = sum;
  for (int m = 1; m < 64; m <<= 1) {
    const int o = __shfl_up(incl, m);
    if ((tid & 63) >= m) incl += o;
  }
  __shared__ int wsum[16];
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  int ex = base + incl - sum;
  int *h = hist + (size_t)b * kSortCells;
  h[c0] = ex;
  h[c0 + 1] = ex + v0;
  h[c0 + 2] = ex + v0 + v1;
  h[c0 + 3] = ex + v0 + v1 + v2;
}

// scatter: perm[sorted position] = original index (order inside a cell is irrelevant)
__global__ __launch_bounds__(256) void cloud_sort_scatter_kernel(int n, const int *__restrict__ cell_of,
                                                               int *__restrict__ hist,
                                                               int *__restrict__ perm, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / n;
    const int k = (int)(e - b * n);
    const int pos = atomicAdd(&hist[b * kSortCells + cell_of[e]], 1);
    perm[b * n + pos] = k;
  }
}

}  // namespace
