// cloud_sort.hpp -- counting sort of a point cloud into the 16^3 cells of a Hilbert curve (gfx950 only).
// Shared by MDS (cluster-sorted slots) and EMD (spatially ordered target stream, seeds).
// One launch per batch of clouds (cloud_sort): one workgroup per cloud computes the bounding box, the cell of
// every point, the exclusive scan of the 4096 cell counts and -- with the counters still in LDS as cursors -- the
// permutation perm[sorted position] = original index; afterwards hist holds the cell END offsets (the order
// inside a cell is arbitrary).  The scatter used to be a second, grid-wide launch with one global atomic per
// point (23 us next to the count kernel's 24 at 32 x 16384 points; together now 30).
#pragma once
#include "common.hpp"

// 16^3 cells.  32^3 (SN_SORT_BITS=5, 128 KB of counters) was measured: the blocks get no tighter for the
// pruning tests at 16384 points and the sort itself costs more (Chamfer 0.58 -> 0.64 ms).
#ifndef SN_SORT_BITS
#define SN_SORT_BITS 4
#endif

namespace {

constexpr int kSortBits = SN_SORT_BITS;          // grid side 2^bits per axis
constexpr int kSortSide = 1 << kSortBits;
constexpr int kSortCells = kSortSide * kSortSide * kSortSide;

// grid coordinate of v inside [lo, lo + ext]: the ONE expression every kernel uses, so a point lands in the
// same cell wherever its cell is recomputed
__device__ __forceinline__ float sort_scale(float ext) { return ext > 0.f ? ((float)kSortSide - 0.001f) / ext : 0.f; }
__device__ __forceinline__ unsigned sort_coord(float v, float lo, float scale) {
  const float f = (v - lo) * scale;
  return (unsigned)(f < 0.f ? 0.f : (f > (float)(kSortSide - 1) ? (float)(kSortSide - 1) : f));
}

// Cell code along a 3-D Hilbert curve over the 2^bits-per-axis grid (Skilling's axes-to-transpose form, then the
// bits interleaved): consecutive codes are face-adjacent cells, so a run of sorted points -- an MDS slot,
// an EMD / Chamfer block of 16 or 64 -- is one compact blob.  Z order jumps at every octant boundary,
// which stretched the bounding boxes the pruning tests use.  (The name is kept: every caller only needs
// "the sort key of a cell".)
__device__ __forceinline__ unsigned morton3_4bit(unsigned x, unsigned y, unsigned z) {
#ifdef SN_SORT_Z_ORDER
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < kSortBits; ++i)
    r |= (((x >> i) & 1u) << (3 * i)) | (((y >> i) & 1u) << (3 * i + 1)) | (((z >> i) & 1u) << (3 * i + 2));
  return r;
#else
  unsigned X[3] = {x, y, z};
#pragma unroll
  for (unsigned q = 1u << (kSortBits - 1); q > 1u; q >>= 1) {
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
  for (unsigned q = 1u << (kSortBits - 1); q > 1u; q >>= 1)
    if (X[2] & q) t ^= q - 1u;
  X[0] ^= t;
  X[1] ^= t;
  X[2] ^= t;
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < kSortBits; ++i)
    r |= (((X[0] >> i) & 1u) << (3 * i + 2)) | (((X[1] >> i) & 1u) << (3 * i + 1)) | (((X[2] >> i) & 1u) << (3 * i));
  return r;
#endif
}

// per cloud: bounding box -> cell histogram -> Hilbert-order permutation (one workgroup per cloud)
struct SortSide {
  int n;
  const float *xyz;
  float *bbox;
  int *hist, *cell_of, *perm;
};

__device__ __forceinline__ void cloud_sort_body(int n, const float *__restrict__ xyz, float *__restrict__ bbox,
                                                int *__restrict__ hist, int *__restrict__ cell_of,
                                                int *__restrict__ perm, int *lh) {
  __shared__ float red[6][16];
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
    scale[a] = sort_scale(h - l);
    if (tid == 0) {
      bbox[b * 6 + a] = l;
      bbox[b * 6 + 3 + a] = h;
    }
  }
  for (int k = tid; k < n; k += 1024) {
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = sort_coord(p[k * 3 + a], blo[a], scale[a]);
    const int c = (int)morton3_4bit(q[0], q[1], q[2]);
    cell_of[(size_t)b * n + k] = c;
    atomicAdd(&lh[c], 1);
  }
  __syncthreads();
  // exclusive scan of the cell counts (kSortCells / 1024 consecutive cells per thread) -> start offsets
  constexpr int kPer = kSortCells / 1024;
  const int c0 = tid * kPer;
  int sum = 0;
  for (int i = 0; i < kPer; ++i) sum += lh[c0 + i];
  int incl = sum;
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
  for (int i = 0; i < kPer; ++i) {  // counts -> start offsets, kept in LDS as the scatter's cursors
    const int cnt = lh[c0 + i];
    lh[c0 + i] = ex;
    ex += cnt;
  }
  __syncthreads();
  for (int k = tid; k < n; k += 1024) {  // the thread that stored cell_of[k] reads it back
    const int pos = atomicAdd(&lh[cell_of[(size_t)b * n + k]], 1);
    perm[(size_t)b * n + pos] = k;
  }
  __syncthreads();
  for (int i = 0; i < kPer; ++i) h[c0 + i] = lh[c0 + i];  // cell END offsets
}

__global__ __launch_bounds__(1024) void cloud_sort_count_kernel(int n, const float *__restrict__ xyz,
                                                              float *__restrict__ bbox,
                                                              int *__restrict__ hist,
                                                              int *__restrict__ cell_of,
                                                              int *__restrict__ perm) {
  extern __shared__ int lh[];  // kSortCells counters
  cloud_sort_body(n, xyz, bbox, hist, cell_of, perm, lh);
}

// two independent sorts (the two clouds of a Chamfer / EMD call) in ONE launch: blockIdx.y picks the side.  A sort
// is one workgroup per cloud for ~45 us whatever the batch -- one after the other they cost twice that, and at 4
// clouds per rank the four sorts of a step were 8 % of it.
__global__ __launch_bounds__(1024) void cloud_sort_pair_kernel(SortSide a, SortSide c) {
  extern __shared__ int lh[];
  const SortSide s = blockIdx.y == 0 ? a : c;
  cloud_sort_body(s.n, s.xyz, s.bbox, s.hist, s.cell_of, s.perm, lh);
}

inline int cloud_sort_pair(int b, const SortSide &a, const SortSide &c, hipStream_t s) {
  if (kSortCells * 4 > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(cloud_sort_pair_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, kSortCells * 4) != hipSuccess)
    return 1;
  cloud_sort_pair_kernel<<<dim3(b, 2), 1024, kSortCells * 4, s>>>(a, c);
  return 0;
}

// the launch: the counters are dynamic LDS (above the 64 KB default at 32^3 cells)
inline int cloud_sort(int b, int n, const float *xyz, float *bbox, int *hist, int *cell_of, int *perm,
                      hipStream_t s) {
  if (kSortCells * 4 > 48 * 1024 &&  // per call: the attribute belongs to the current device
      hipFuncSetAttribute(reinterpret_cast<const void *>(cloud_sort_count_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, kSortCells * 4) != hipSuccess)
    return 1;
  cloud_sort_count_kernel<<<b, 1024, kSortCells * 4, s>>>(n, xyz, bbox, hist, cell_of, perm);
  return 0;
}

}  // namespace
