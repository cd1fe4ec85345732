// chamfer.hip -- Chamfer distance forward / backward for MI355X (gfx950).
//
// Reference semantics: cuda/chamfer_distance/chamfer_distance.cu:7-137 (forward),
// :159-209 (backward); CPU statement chamfer_distance.cpp:57-180.
//   dist[b,j] = min_k d(j,k),  d = (dx*dx + dy*dy) + dz*dz with dx = t.x - q.x,
//   three separately rounded products and two rounded sums (no FMA),
//   idx[b,j]  = LOWEST k attaining the minimum.
//
// Design (wave64, fp32-VALU bound -- 9 flop/pair, O(N) bytes):
//   * one lane owns QPL=4 queries in VGPRs, packed two-by-two so the distance
//     arithmetic issues as v_pk_add_f32 / v_pk_mul_f32 (2 pairs per VALU slot);
//   * targets stream through a 1024-point LDS tile laid out chunk-SoA
//     (x[8] y[8] z[8] per chunk) and are read with wave-uniform ds_read_b128
//     (broadcast, conflict free); the next tile's global loads are issued
//     before the current tile is consumed;
//   * arg-min is tracked per 8-target CHUNK: the chunk minimum is a v_min3_f32
//     tree (0.5 op/pair) and only one compare + two selects per chunk touch the
//     running (best, best_chunk).  Minimum is exact in fp32, so "first chunk
//     whose minimum beats the running best with strict <" followed by "first k
//     inside that chunk whose recomputed distance equals the minimum" is exactly
//     "lowest k attaining the minimum";
//   * both directions (1->2 and 2->1) run in ONE launch; blocks that share a
//     target cloud are mapped to the same XCD (block id % 8) so the cloud's
//     196 KB stay in one L2.
#include "common.hpp"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kQPL = 4;                  // queries per lane
constexpr int kQPB = kThreads * kQPL;    // queries per block
constexpr int kChunk = 8;                // targets per arg-min chunk
constexpr int kTile = 1024;              // targets per LDS tile
constexpr int kTileF4 = kTile / kChunk * 6;  // float4 slots per tile

__device__ __forceinline__ float min3(float a, float b, float c) {
  return __builtin_fminf(__builtin_fminf(a, b), c);
}

// d = (dx*dx + dy*dy) + dz*dz, two queries at once, no contraction
__device__ __forceinline__ f2 dist2(float tx, float ty, float tz, f2 qx, f2 qy, f2 qz) {
#pragma clang fp contract(off)
  const f2 dx = tx - qx;
  const f2 dy = ty - qy;
  const f2 dz = tz - qz;
  const f2 xx = dx * dx;
  const f2 yy = dy * dy;
  const f2 zz = dz * dz;
  const f2 s = xx + yy;
  return s + zz;
}

__device__ __forceinline__ float dist1(float tx, float ty, float tz, float qx, float qy, float qz) {
#pragma clang fp contract(off)
  const float dx = tx - qx;
  const float dy = ty - qy;
  const float dz = tz - qz;
  const float xx = dx * dx;
  const float yy = dy * dy;
  const float zz = dz * dz;
  const float s = xx + yy;
  return s + zz;
}

__global__ __launch_bounds__(kThreads) void chamfer_fwd_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, int B, int N, int M,
    float *__restrict__ dist1_out, int *__restrict__ idx1_out,
    float *__restrict__ dist2_out, int *__restrict__ idx2_out, int nb1, int nb2, int nb_max) {
  __shared__ float4 tile[kTileF4];

  // ---- XCD-aware decode: block g runs on XCD g%8; all blocks of one target
  // cloud (b, direction) share that residue.
  const int g = blockIdx.x;
  const int xcd = g & 7;
  const int r = g >> 3;
  const int cloud = (r / nb_max) * 8 + xcd;
  const int blk = r % nb_max;
  if (cloud >= 2 * B) return;
  const int b = cloud >> 1;
  const int dir = cloud & 1;
  if (blk >= (dir ? nb2 : nb1)) return;

  const int nq = dir ? M : N;   // queries
  const int nt = dir ? N : M;   // targets
  const float *__restrict__ q = (dir ? xyz2 : xyz1) + (size_t)b * nq * 3;
  const float *__restrict__ t = (dir ? xyz1 : xyz2) + (size_t)b * nt * 3;
  float *__restrict__ dist_out = (dir ? dist2_out : dist1_out) + (size_t)b * nq;
  int *__restrict__ idx_out = (dir ? idx2_out : idx1_out) + (size_t)b * nq;

  const int tid = threadIdx.x;
  const int q0 = blk * kQPB + tid;

  f2 qx[kQPL / 2], qy[kQPL / 2], qz[kQPL / 2];
#pragma unroll
  for (int i = 0; i < kQPL; ++i) {
    int j = q0 + i * kThreads;
    j = j < nq ? j : nq - 1;
    qx[i >> 1][i & 1] = q[j * 3 + 0];
    qy[i >> 1][i & 1] = q[j * 3 + 1];
    qz[i >> 1][i & 1] = q[j * 3 + 2];
  }

  float best[kQPL];
  int bchunk[kQPL];
#pragma unroll
  for (int i = 0; i < kQPL; ++i) {
    best[i] = __builtin_inff();
    bchunk[i] = 0;
  }

  const int ntiles = sn::ceil_div(nt, kTile);
  constexpr int kLd = kTile * 3 / kThreads;  // floats per thread per tile (12)
  float stage[kLd];
  const int nt3 = nt * 3;

  auto load_stage = [&](int tile_id) {
    const int base = tile_id * kTile * 3;
#pragma unroll
    for (int i = 0; i < kLd; ++i) {
      const int e = base + i * kThreads + tid;
      stage[i] = e < nt3 ? t[e] : __builtin_inff();
    }
  };
  auto store_stage = [&]() {
    float *lds = reinterpret_cast<float *>(tile);
#pragma unroll
    for (int i = 0; i < kLd; ++i) {
      const int e = i * kThreads + tid;
      const int k = e / 3, comp = e - k * 3;
      lds[(k >> 3) * 24 + comp * 8 + (k & 7)] = stage[i];
    }
  };

  load_stage(0);
  for (int tile_id = 0; tile_id < ntiles; ++tile_id) {
    __syncthreads();  // previous tile fully consumed
    store_stage();
    __syncthreads();
    if (tile_id + 1 < ntiles) load_stage(tile_id + 1);

    const int chunk0 = tile_id * (kTile / kChunk);
    const int rem = nt - tile_id * kTile;
    const int nchunks = rem >= kTile ? kTile / kChunk : sn::ceil_div(rem, kChunk);
#pragma unroll 2
    for (int c = 0; c < nchunks; ++c) {
      const float4 xa = tile[c * 6 + 0], xb = tile[c * 6 + 1];
      const float4 ya = tile[c * 6 + 2], yb = tile[c * 6 + 3];
      const float4 za = tile[c * 6 + 4], zb = tile[c * 6 + 5];
      const float tx[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      const float ty[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
      const float tz[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
      for (int p = 0; p < kQPL / 2; ++p) {
        f2 d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = dist2(tx[k], ty[k], tz[k], qx[p], qy[p], qz[p]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float m = min3(d[0][h], d[1][h], d[2][h]);
          m = min3(m, d[3][h], d[4][h]);
          m = min3(m, d[5][h], d[6][h]);
          m = __builtin_fminf(m, d[7][h]);
          const int i = p * 2 + h;
          const bool lt = m < best[i];
          best[i] = lt ? m : best[i];
          bchunk[i] = lt ? chunk0 + c : bchunk[i];
        }
      }
    }
  }

  // ---- epilogue: first k inside the winning chunk whose distance equals best
#pragma unroll
  for (int i = 0; i < kQPL; ++i) {
    const int j = q0 + i * kThreads;
    if (j >= nq) continue;
    const float x = qx[i >> 1][i & 1], y = qy[i >> 1][i & 1], z = qz[i >> 1][i & 1];
    const int k0 = bchunk[i] * kChunk;
    int bi = k0;
    bool found = false;
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {
      const int kk = k0 + k < nt ? k0 + k : nt - 1;
      const float d = dist1(t[kk * 3 + 0], t[kk * 3 + 1], t[kk * 3 + 2], x, y, z);
      const bool hit = !found && (d == best[i]) && (k0 + k < nt);
      bi = hit ? k0 + k : bi;
      found = found || hit;
    }
    dist_out[j] = best[i];
    idx_out[j] = bi;
  }
}

// ---------------------------------------------------------------- backward
// The reference's CPU path (chamfer_distance.cpp:114-180) accumulates, per cloud, first over the queries of
// cloud 1 in ascending order (own term into gradxyz1, scatter term into gradxyz2[idx1[j]]), then over the
// queries of cloud 2 (own term into gradxyz2, scatter term into gradxyz1[idx2[k]]):
//     gradxyz1[j] = ( own1_j - s_{k1} - s_{k2} ... )            k ascending over {k : idx2[k] == j}
//     gradxyz2[k] = ( -t_{j1} - t_{j2} ... ) + own2_k           j ascending over {j : idx1[j] == k}
// Its GPU path scatters with fp32 atomics -- the same terms in an arbitrary order, so the last bits depend on
// timing.  Here the scatter is turned into a GATHER: the inverse lists (who points at me?) are built with
// integer atomics (count -> scan -> fill; the fill order is arbitrary, so every list is sorted before it is
// used) and each point then adds its terms in exactly the order above: no floating-point atomic, results
// bit-reproducible and bit-equal to the reference's own CPU build.
// a neighbour index from the caller (cd.backward_cuda takes idx tensors): out-of-range values would address LDS /
// the lists out of bounds, so they are clamped (the forward never produces one)
__device__ __forceinline__ int clamp_idx(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

constexpr int kLongList = 64;      // inverse lists longer than this are sorted and summed by a workgroup
constexpr int kLongSortCap = 16384;  // ... in LDS up to this length (64 KB); longer ones by one thread, in place

struct BwdLists {
  int *cnt;    // [B, N + M]  how many points of the other cloud chose me (entries 0..N-1: cloud 1, then cloud 2)
  int *off;    // [B, N + M]  list start
  int *fill;   // [B, N + M]  next free slot while filling
  int *list;   // [B, N + M]  the inverse lists: lists of cloud-1 points hold indices k of cloud 2 and vice versa
  int *longs;  // [1 + B (N + M) / 64]  count, then the global slots (b (N + M) + slot) of the lists > kLongList
};

__global__ __launch_bounds__(256) void chamfer_bwd_count_kernel(const int *__restrict__ idx1,
                                                               const int *__restrict__ idx2, int B, int N,
                                                               int M, int *__restrict__ cnt) {
  const long total1 = (long)B * N, total = total1 + (long)B * M;
  const int NM = N + M;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const bool second = e >= total1;
    const long f = second ? e - total1 : e;
    const int b = (int)(f / (second ? M : N));
    // a cloud-1 query j votes for cloud-2 point idx1[j] (slot N + idx1[j]); a cloud-2 query for slot idx2[k]
    const int slot = second ? clamp_idx(idx2[f], N) : N + clamp_idx(idx1[f], M);
    atomicAdd(&cnt[(long)b * NM + slot], 1);
  }
}

// exclusive scan of the counts of one cloud (both sides in one sequence); also primes the fill cursors
__global__ __launch_bounds__(1024) void chamfer_bwd_scan_kernel(int NM, const int *__restrict__ cnt,
                                                               int *__restrict__ off, int *__restrict__ fill,
                                                               int *__restrict__ longs) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long o = (long)b * NM;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < NM; base += 1024) {
    const int i = base + tid;
    const int c = i < NM ? cnt[o + i] : 0;
    int incl = c;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < (tid >> 6); ++w) pre += wsum[w];
    if (i < NM) {
      off[o + i] = pre + incl - c;
      fill[o + i] = pre + incl - c;
      if (c > kLongList) longs[1 + atomicAdd(&longs[0], 1)] = (int)(o + i);
    }
    __syncthreads();
    if (tid == 1023) carry = pre + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void chamfer_bwd_fill_kernel(const int *__restrict__ idx1,
                                                              const int *__restrict__ idx2, int B, int N,
                                                              int M, int *__restrict__ fill,
                                                              int *__restrict__ list) {
  const long total1 = (long)B * N, total = total1 + (long)B * M;
  const int NM = N + M;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const bool second = e >= total1;
    const long f = second ? e - total1 : e;
    const int na = second ? M : N;
    const int b = (int)(f / na);
    const int self = (int)(f - (long)b * na);
    const int slot = second ? clamp_idx(idx2[f], N) : N + clamp_idx(idx1[f], M);
    const int pos = atomicAdd(&fill[(long)b * NM + slot], 1);
    list[(long)b * NM + pos] = self;
  }
}

// count + scan + fill of one cloud in ONE workgroup with the N + M counters in LDS (<= 36864 of them): no global
// atomics, no counter round trips through memory (the three kernels above: 0.14 ms at B = 32, N = M = 16384; this
// one 0.03).  Leaves cnt / off / list exactly as they do (the order inside a list is arbitrary either way: the
// gather sorts every list).
constexpr int kBwdLdsSlots = 36864;
__global__ __launch_bounds__(1024) void chamfer_bwd_lists_kernel(const int *__restrict__ idx1,
                                                               const int *__restrict__ idx2, int N, int M,
                                                               int *__restrict__ cnt, int *__restrict__ off,
                                                               int *__restrict__ list, int *__restrict__ longs) {
  extern __shared__ int lc[];  // N + M counters, later the fill cursors
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x, NM = N + M;
  const long o = (long)b * NM;
  const int *i1 = idx1 + (long)b * N, *i2 = idx2 + (long)b * M;
  for (int i = tid; i < NM; i += 1024) lc[i] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int e = tid; e < NM; e += 1024) atomicAdd(&lc[e >= N ? clamp_idx(i2[e - N], N) : N + clamp_idx(i1[e], M)], 1);
  __syncthreads();
  for (int base = 0; base < NM; base += 1024) {
    const int i = base + tid;
    const int c = i < NM ? lc[i] : 0;
    int incl = c;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < (tid >> 6); ++w) pre += wsum[w];
    if (i < NM) {
      cnt[o + i] = c;
      off[o + i] = pre + incl - c;
      lc[i] = pre + incl - c;
      if (c > kLongList) longs[1 + atomicAdd(&longs[0], 1)] = (int)(o + i);
    }
    __syncthreads();
    if (tid == 1023) carry = pre + incl;
    __syncthreads();
  }
  for (int e = tid; e < NM; e += 1024) {
    const bool second = e >= N;
    const int pos = atomicAdd(&lc[second ? clamp_idx(i2[e - N], N) : N + clamp_idx(i1[e], M)], 1);
    list[o + pos] = second ? e - N : e;
  }
}

// in-place ascending sort of a short int array in global memory owned by the calling thread
__device__ __forceinline__ void sort_ints(int *a, int n) {
  if (n <= 16) {  // insertion sort
    for (int i = 1; i < n; ++i) {
      const int v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) {
        a[j + 1] = a[j];
        --j;
      }
      a[j + 1] = v;
    }
    return;
  }
  // heap sort: O(n log n) also for the degenerate case of thousands of queries sharing one neighbour
  auto sift = [&](int start, int end) {
    int root = start;
    for (;;) {
      int child = 2 * root + 1;
      if (child > end) break;
      if (child + 1 <= end && a[child] < a[child + 1]) ++child;
      if (a[root] >= a[child]) break;
      const int t = a[root];
      a[root] = a[child];
      a[child] = t;
      root = child;
    }
  };
  for (int start = (n - 2) / 2; start >= 0; --start) sift(start, n - 1);
  for (int end = n - 1; end > 0; --end) {
    const int t = a[0];
    a[0] = a[end];
    a[end] = t;
    sift(0, end - 1);
  }
}

__global__ __launch_bounds__(256) void chamfer_bwd_gather_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, const float *__restrict__ gd1,
    const float *__restrict__ gd2, const int *__restrict__ idx1, const int *__restrict__ idx2, int B, int N,
    int M, const int *__restrict__ cnt, const int *__restrict__ off, int *__restrict__ list,
    float *__restrict__ g1, float *__restrict__ g2) {
#pragma clang fp contract(off)
  const long total1 = (long)B * N, total = total1 + (long)B * M;
  const int NM = N + M;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const bool second = e >= total1;
    const long f = second ? e - total1 : e;
    const int na = second ? M : N, nb = second ? N : M;
    const int b = (int)(f / na);
    const int self = (int)(f - (long)b * na);
    const float *a = second ? xyz2 : xyz1;   // my cloud
    const float *o = second ? xyz1 : xyz2;   // the other cloud
    const float *gda = second ? gd2 : gd1, *gdo = second ? gd1 : gd2;
    const float *pa = a + f * 3;
    // own term: g (me - my neighbour)
    const int k = clamp_idx((second ? idx2 : idx1)[f], nb);
    const float *po = o + ((long)b * nb + k) * 3;
    const float g = gda[f] * 2;
    const float own[3] = {g * (pa[0] - po[0]), g * (pa[1] - po[1]), g * (pa[2] - po[2])};
    const long slot = (long)b * NM + (second ? N : 0) + self;
    const int c = cnt[slot];
    if (c > kLongList) continue;  // chamfer_bwd_long_kernel's
    int *lst = list + (long)b * NM + off[slot];
    if (c > 1) sort_ints(lst, c);
    // cloud 1: own term first, then the scatter terms; cloud 2: scatter terms first, own term last
    float acc[3] = {second ? 0.f : own[0], second ? 0.f : own[1], second ? 0.f : own[2]};
    for (int i = 0; i < c; ++i) {
      const long q = (long)b * nb + lst[i];            // a point of the other cloud whose neighbour I am
      const float *pq = o + q * 3;
      const float gq = gdo[q] * 2;
      acc[0] -= gq * (pq[0] - pa[0]);
      acc[1] -= gq * (pq[1] - pa[1]);
      acc[2] -= gq * (pq[2] - pa[2]);
    }
    float *out = (second ? g2 : g1) + f * 3;
    out[0] = second ? acc[0] + own[0] : acc[0];
    out[1] = second ? acc[1] + own[1] : acc[1];
    out[2] = second ? acc[2] + own[2] : acc[2];
  }
}

// Long inverse lists: early in training (tanh outputs near 0) thousands of queries can share ONE nearest neighbour.
// The gather above would let a single thread sort such a list in global memory and add its terms one dependent
// load at a time (milliseconds for a 16384-entry list, where the reference's atomic scatter needs microseconds).
// Here a workgroup takes the list: bitonic sort in LDS, then the terms in ascending order -- products computed 64 at
// a time by a wave, the three running sums kept by three lanes (the order of the additions is the CPU path's, so
// the result stays bit-equal to it).
__global__ __launch_bounds__(256) void chamfer_bwd_long_kernel(
    const float *__restrict__ xyz1, const float *__restrict__ xyz2, const float *__restrict__ gd1,
    const float *__restrict__ gd2, const int *__restrict__ idx1, const int *__restrict__ idx2, int B, int N,
    int M, const int *__restrict__ cnt, const int *__restrict__ off, int *__restrict__ list,
    const int *__restrict__ longs, float *__restrict__ g1, float *__restrict__ g2) {
#pragma clang fp contract(off)
  extern __shared__ int keys[];  // kLongSortCap ints
  __shared__ float terms[3][64];
  const int tid = threadIdx.x, NM = N + M;
  const int nlong = longs[0];
  for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
    const long gslot = longs[1 + li];
    const int b = (int)(gslot / NM), sl = (int)(gslot - (long)b * NM);
    const bool second = sl >= N;
    const int self = second ? sl - N : sl;
    const int nb = second ? N : M;
    const long f = (long)b * (second ? M : N) + self;
    const float *a = second ? xyz2 : xyz1, *o = second ? xyz1 : xyz2;
    const float *gda = second ? gd2 : gd1, *gdo = second ? gd1 : gd2;
    const float *pa = a + f * 3;
    const int c = cnt[gslot];
    int *lst = list + (long)b * NM + off[gslot];
    __syncthreads();  // the previous list's LDS keys are no longer needed
    if (c <= kLongSortCap) {
      int p2 = 1;
      while (p2 < c) p2 <<= 1;
      for (int i = tid; i < p2; i += 256) keys[i] = i < c ? lst[i] : 0x7fffffff;
      __syncthreads();
      for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < p2; i += 256) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const int x = keys[i], y = keys[ixj];
              const bool up = (i & k) == 0;
              if ((x > y) == up) {
                keys[i] = y;
                keys[ixj] = x;
              }
            }
          }
          __syncthreads();
        }
    } else {
      if (tid == 0) sort_ints(lst, c);
      __threadfence_block();
      __syncthreads();
    }
    if (tid < 64) {  // wave 0: ordered accumulation
      const int k = clamp_idx((second ? idx2 : idx1)[f], nb);
      const float *po = o + ((long)b * nb + k) * 3;
      const float g = gda[f] * 2;
      const float ax = tid < 3 ? pa[tid] : 0.f;
      const float own = tid < 3 ? g * (ax - po[tid]) : 0.f;
      float acc = second ? 0.f : own;   // lanes 0..2: x, y, z
      for (int base = 0; base < c; base += 64) {
        const int i = base + tid;
        if (i < c) {
          const long q = (long)b * nb + (c <= kLongSortCap ? keys[i] : lst[i]);
          const float *pq = o + q * 3;
          const float gq = gdo[q] * 2;
          terms[0][tid] = gq * (pq[0] - pa[0]);
          terms[1][tid] = gq * (pq[1] - pa[1]);
          terms[2][tid] = gq * (pq[2] - pa[2]);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int lim = c - base < 64 ? c - base : 64;
        if (tid < 3)
          for (int t = 0; t < lim; ++t) acc -= terms[tid][t];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (tid < 3) (second ? g2 : g1)[f * 3 + tid] = second ? acc + own : acc;
    }
  }
}

}  // namespace

extern "C" int sn_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n, int m,
                                  float *dist1, int *idx1, float *dist2, int *idx2,
                                  void *stream) {
  SN_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2, "sn_chamfer_forward: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1 && m >= 1, "sn_chamfer_forward: need b,n,m >= 1 (got %d,%d,%d)", b, n, m);
  SN_REQUIRE((long)b * n < (1L << 29) && (long)b * m < (1L << 29), "sn_chamfer_forward: too large");
  const int nb1 = sn::ceil_div(n, kQPB), nb2 = sn::ceil_div(m, kQPB);
  const int nb_max = nb1 > nb2 ? nb1 : nb2;
  const int clouds_per_xcd = sn::ceil_div(2 * b, 8);
  const int grid = 8 * clouds_per_xcd * nb_max;
  hipStream_t s = sn::as_stream(stream);
  SN_TIMED("chamfer_fwd", s, (chamfer_fwd_kernel<<<grid, kThreads, 0, s>>>(
      xyz1, xyz2, b, n, m, dist1, idx1, dist2, idx2, nb1, nb2, nb_max)));
  return sn::launch_status("sn_chamfer_forward");
}

extern "C" size_t sn_chamfer_backward_workspace_bytes(int b, int n, int m) {
  if (b < 1 || n < 1 || m < 1) return 0;
  const size_t arr = sn::align_up((size_t)b * ((size_t)n + m) * 4, 256);
  return 4 * arr + sn::align_up(arr / 64 + 256, 256);  // cnt, off, fill, list + the directory of the long lists
}

extern "C" int sn_chamfer_backward(const float *xyz1, const float *xyz2, const float *graddist1,
                                   const float *graddist2, const int *idx1, const int *idx2,
                                   int b, int n, int m, float *gradxyz1, float *gradxyz2,
                                   void *workspace, size_t workspace_bytes, void *stream) {
  SN_REQUIRE(xyz1 && xyz2 && graddist1 && graddist2 && idx1 && idx2 && gradxyz1 && gradxyz2 && workspace,
             "sn_chamfer_backward: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1 && m >= 1, "sn_chamfer_backward: need b,n,m >= 1");
  SN_REQUIRE((long)b * ((long)n + m) < (1L << 30), "sn_chamfer_backward: too large");
  SN_REQUIRE(workspace_bytes >= sn_chamfer_backward_workspace_bytes(b, n, m),
             "sn_chamfer_backward: workspace too small (%zu < %zu)", workspace_bytes,
             sn_chamfer_backward_workspace_bytes(b, n, m));
  const size_t arr = sn::align_up((size_t)b * ((size_t)n + m) * 4, 256);
  char *p = static_cast<char *>(workspace);
  BwdLists L;
  L.cnt = reinterpret_cast<int *>(p);
  L.off = reinterpret_cast<int *>(p + arr);
  L.fill = reinterpret_cast<int *>(p + 2 * arr);
  L.list = reinterpret_cast<int *>(p + 3 * arr);
  L.longs = reinterpret_cast<int *>(p + 4 * arr);
  const long total = (long)b * n + (long)b * m;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = sn::as_stream(stream);
  SN_REFUSE_CAPTURE(s, "sn_chamfer_backward");
  SN_HIP(hipMemsetAsync(L.longs, 0, 4, s));
  const size_t lds = ((size_t)n + m) * 4;
  if (n + m <= kBwdLdsSlots &&
      (lds <= 48 * 1024 ||  // per call: the attribute belongs to the current device
       hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_bwd_lists_kernel),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)) {
    chamfer_bwd_lists_kernel<<<b, 1024, lds, s>>>(idx1, idx2, n, m, L.cnt, L.off, L.list, L.longs);
  } else {
    SN_HIP(hipMemsetAsync(L.cnt, 0, (size_t)b * ((size_t)n + m) * 4, s));
    chamfer_bwd_count_kernel<<<(int)blocks, 256, 0, s>>>(idx1, idx2, b, n, m, L.cnt);
    chamfer_bwd_scan_kernel<<<b, 1024, 0, s>>>(n + m, L.cnt, L.off, L.fill, L.longs);
    chamfer_bwd_fill_kernel<<<(int)blocks, 256, 0, s>>>(idx1, idx2, b, n, m, L.fill, L.list);
  }
  chamfer_bwd_gather_kernel<<<(int)blocks, 256, 0, s>>>(xyz1, xyz2, graddist1, graddist2, idx1, idx2, b, n, m,
                                                        L.cnt, L.off, L.list, gradxyz1, gradxyz2);
  // the lists beyond kLongList entries (none in a trained model: the kernel then reads one word and leaves)
  SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_bwd_long_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, kLongSortCap * 4));
  chamfer_bwd_long_kernel<<<128, 256, kLongSortCap * 4, s>>>(xyz1, xyz2, graddist1, graddist2, idx1, idx2, b, n, m,
                                                            L.cnt, L.off, L.list, L.longs, gradxyz1, gradxyz2);
  return sn::launch_status("sn_chamfer_backward");
}
