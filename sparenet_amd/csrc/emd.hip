// emd.hip -- auction-based approximate Earth Mover's Distance for MI355X (gfx950).
//
// Reference: cuda/emd/emd_cuda.cu:23-282 (forward: 7 launches per iteration + 1),
// :284-316 (backward); scratch tensors of cuda/emd/emd_module.py:43-54.
//
// Semantics kept bit-exactly (see oracle/emd.c for the sequential statement):
//   bid value  d = (float)((3.0 - (double)sqrtf(s)) - (double)price[k]),
//              s = (dx*dx + dy*dy) + dz*dz, separately rounded, dx = xyz2 - xyz1;
//   per bidder best = max d, better = second max (duplicates count, init -1e9);
//   exact ties at the top resolve to argmin (thread(k), k), thread(k) being the
//   reference's chunk of the 2048-tile (emd_cuda.cu:136-139) -- handled by a rare
//   re-scan, because the top-2 VALUES are partition independent;
//   GetMax window +-1e-6 in double; several bidders inside the window -> highest
//   bidder index (what a sequential ascending-j run of :188-191 gives; a race on
//   the reference's GPU); Assign/eviction/price update/last-iteration force
//   assignment exactly as :196-215.
//
// MI355X design
//   * 3 launches per iteration instead of 7: the unassigned list for the next
//     iteration is produced by Assign itself (losers and evicted points append
//     with a wave-aggregated atomic) -- no count / scan / compaction kernels.
//   * Bid: the unassigned count U_b is only known on the device, so a fixed grid
//     (G blocks per cloud) adapts: T = lanes per bidder = 2^k <= 64 chosen from
//     U_b so that all G*256 lanes of a cloud are busy; each lane scans the
//     targets t, t+T, ... of every LDS tile (SoA x|y|z|price, conflict-free for
//     any T, pure broadcast for T=1) and the T partial top-2's merge with a
//     wave64 xor-butterfly (no LDS, no barrier).  Tail iterations with a
//     handful of bidders therefore still spread over whole waves.
//   * GetMax: deterministic atomicMax of the bidder index inside the window.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;      // targets per LDS tile (SoA x,y,z,price = 16 KB)
constexpr int kBlocksPerCloud = 64;

struct Top2 {
  float best, better;
  int best_i;
};

__device__ __forceinline__ float bid_value(float tx, float ty, float tz, float p, float x1,
                                           float y1, float z1) {
#pragma clang fp contract(off)
  const float dx = tx - x1, dy = ty - y1, dz = tz - z1;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const float s = (xx + yy) + zz;
  return (float)((3.0 - (double)__builtin_sqrtf(s)) - (double)p);
}

__device__ __forceinline__ void top2_push(Top2 &t, float d, int k) {
  // if (d > best) {better = best; best = d; best_i = k} else if (d > better) better = d
  const bool gt = d > t.best;
  t.better = __builtin_fmaxf(t.better, __builtin_fminf(d, t.best));
  t.best_i = gt ? k : t.best_i;
  t.best = gt ? d : t.best;
}

__device__ __forceinline__ void top2_merge(Top2 &a, float b_best, float b_better, int b_i) {
  const bool gt = b_best > a.best;
  const float lo = gt ? a.best : b_best;
  a.better = __builtin_fmaxf(__builtin_fmaxf(a.better, b_better), lo);
  a.best_i = gt ? b_i : a.best_i;
  a.best = gt ? b_best : a.best;
}

// order-preserving float max through integer atomics
__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
  if (v >= 0.f)
    atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

struct EmdWs {
  int *assignment_inv;
  float *price;
  int *bid;
  float *bid_inc;
  float *max_inc;
  int *max_idx;
  int *list[2];
  int *cnt[2];
};

__global__ void emd_init_kernel(int B, int n, int *__restrict__ assignment, EmdWs ws) {
  const long total = (long)B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    assignment[e] = -1;
    ws.assignment_inv[e] = -1;
    ws.price[e] = 0.f;
    ws.max_inc[e] = 0.f;  // emd_module.py:49 (zeros, not -1e9)
    ws.max_idx[e] = 0;
    ws.list[0][e] = (int)(e % n);
    if (e < B) {
      ws.cnt[0][e] = n;
      ws.cnt[1][e] = 0;
    }
  }
}

// lanes per bidder for a cloud with U bidders: largest 2^k <= lanes/U, in [1,64]
__device__ __forceinline__ int lanes_per_bidder(int U) {
  const int lanes = kBlocksPerCloud * kThreads;
  int T = 1;
  while (T < 64 && T * 2 * U <= lanes) T *= 2;
  return T;
}

__global__ __launch_bounds__(kThreads) void emd_bid_kernel(
    int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2, float eps,
    const float *__restrict__ price, int *__restrict__ bid, float *__restrict__ bid_inc,
    float *__restrict__ max_inc, int *__restrict__ max_idx, const int *__restrict__ list,
    const int *__restrict__ cnt, long long *__restrict__ stats) {
  __shared__ float sx[kTile], sy[kTile], sz[kTile], sp[kTile];
  const int b = blockIdx.y;
  const int U = cnt[b];
  if (U == 0) return;
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd(reinterpret_cast<unsigned long long *>(stats), (unsigned long long)U * n);
    if (b == 0) atomicAdd(reinterpret_cast<unsigned long long *>(stats) + 1, 1ULL);
  }
  const int T = lanes_per_bidder(U);
  const int per_block = kThreads / T;
  const int tid = threadIdx.x;
  const int t = tid & (T - 1);
  const int g = tid / T;
  const float *__restrict__ p1 = xyz1 + (size_t)b * n * 3;
  const float *__restrict__ p2 = xyz2 + (size_t)b * n * 3;
  const float *__restrict__ pr = price + (size_t)b * n;
  const int *__restrict__ lst = list + (size_t)b * n;

  // reference partition, only needed to order exact ties (emd_cuda.cu:108-109,136)
  const int block_cnt = n / 1024;
  const int tpu_ref = 1024 / ((U + block_cnt - 1) / block_cnt);

  for (int u0 = blockIdx.x * per_block; u0 < U; u0 += gridDim.x * per_block) {
    const int u = u0 + g;
    const bool active = u < U;
    const int j = active ? lst[u] : 0;
    const float x1 = p1[j * 3 + 0], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
    Top2 top = {-1e9f, -1e9f, -1};

    for (int k0 = 0; k0 < n; k0 += kTile) {
      __syncthreads();
      for (int k = tid; k < kTile; k += kThreads) {  // n % 1024 == 0: tiles are full
        sx[k] = p2[(k0 + k) * 3 + 0];
        sy[k] = p2[(k0 + k) * 3 + 1];
        sz[k] = p2[(k0 + k) * 3 + 2];
        sp[k] = pr[k0 + k];
      }
      __syncthreads();
#pragma unroll 4
      for (int k = t; k < kTile; k += T) {
        const float d = bid_value(sx[k], sy[k], sz[k], sp[k], x1, y1, z1);
        top2_push(top, d, k0 + k);
      }
    }
    // merge the T partial results (lanes of a bidder are contiguous, T <= 64)
    for (int m = 1; m < T; m <<= 1) {
      const float ob = __shfl_xor(top.best, m);
      const float os = __shfl_xor(top.better, m);
      const int oi = __shfl_xor(top.best_i, m);
      top2_merge(top, ob, os, oi);
    }
    // exact tie at the top: canonical best_i = argmin (thread_ref(k), k)
    const bool tied = active && (top.best == top.better);
    if (__syncthreads_or(tied)) {
      int key = 0x7fffffff;
      for (int k0 = 0; k0 < n; k0 += kTile) {
        __syncthreads();
        for (int k = tid; k < kTile; k += kThreads) {
          sx[k] = p2[(k0 + k) * 3 + 0];
          sy[k] = p2[(k0 + k) * 3 + 1];
          sz[k] = p2[(k0 + k) * 3 + 2];
          sp[k] = pr[k0 + k];
        }
        __syncthreads();
        if (tied) {
          // 2048-tile geometry of the reference for this k0
          const int ref_k2 = (k0 / 2048) * 2048;
          const int end_k = (n < ref_k2 + 2048 ? n : ref_k2 + 2048) - ref_k2;
          const int delta = (end_k + tpu_ref - 1) / tpu_ref;
          for (int k = t; k < kTile; k += T) {
            const float d = bid_value(sx[k], sy[k], sz[k], sp[k], x1, y1, z1);
            if (d == top.best) {
              const int kk = k0 + k;
              const int thr = (kk - ref_k2) / delta;
              const int cand = thr * (1 << 20) + kk;  // n <= 2^20 checked by the host
              key = cand < key ? cand : key;
            }
          }
        }
      }
      for (int m = 1; m < T; m <<= 1) {
        const int ok = __shfl_xor(key, m);
        key = ok < key ? ok : key;
      }
      if (tied) top.best_i = key & ((1 << 20) - 1);
    }
    if (active && t == 0) {
      const float inc = (top.best - top.better) + eps;
      bid[(size_t)b * n + j] = top.best_i;
      bid_inc[(size_t)b * n + j] = inc;
      atomic_max_float(&max_inc[(size_t)b * n + top.best_i], inc);
      max_idx[(size_t)b * n + top.best_i] = -1;  // winner is re-derived by emd_getmax_kernel
    }
  }
}

__global__ __launch_bounds__(kThreads) void emd_getmax_kernel(
    int n, const int *__restrict__ bid, const float *__restrict__ bid_inc,
    const float *__restrict__ max_inc, int *__restrict__ max_idx, const int *__restrict__ list,
    const int *__restrict__ cnt, int *__restrict__ cnt_next) {
  const int b = blockIdx.y;
  const int U = cnt[b];
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt_next[b] = 0;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
    const int j = list[(size_t)b * n + u];
    const int tgt = bid[(size_t)b * n + j];
    const float bi = bid_inc[(size_t)b * n + j];
    const float mi = max_inc[(size_t)b * n + tgt];
    if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6)
      atomicMax(&max_idx[(size_t)b * n + tgt], j);
  }
}

__global__ __launch_bounds__(kThreads) void emd_assign_kernel(
    int n, int *__restrict__ assignment, int *__restrict__ assignment_inv,
    float *__restrict__ price, const int *__restrict__ bid, const float *__restrict__ bid_inc,
    float *__restrict__ max_inc, const int *__restrict__ max_idx, const int *__restrict__ list,
    const int *__restrict__ cnt, int *__restrict__ list_next, int *__restrict__ cnt_next,
    int last) {
  const int b = blockIdx.y;
  const int U = cnt[b];
  const size_t o = (size_t)b * n;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
    const int j = list[o + u];
    const int tgt = bid[o + j];
    if (last || max_idx[o + tgt] == j) {
      const int inv = assignment_inv[o + tgt];
      if (!last && inv != -1) {
        assignment[o + inv] = -1;
        list_next[o + atomicAdd(&cnt_next[b], 1)] = inv;
      }
      assignment_inv[o + tgt] = j;
      assignment[o + j] = tgt;
      price[o + tgt] += bid_inc[o + j];
      max_inc[o + tgt] = -1e9f;
    } else {
      list_next[o + atomicAdd(&cnt_next[b], 1)] = j;
    }
  }
}

__global__ __launch_bounds__(kThreads) void emd_calcdist_kernel(
    int B, int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const int *__restrict__ assignment, float *__restrict__ dist) {
#pragma clang fp contract(off)
  const long total = (long)B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int k = assignment[e];
    if (k < 0) {  // only with iters == 0 (undefined in the reference)
      dist[e] = 0.f;
      continue;
    }
    const long bb = e / n;
    const float *a = xyz1 + e * 3, *o = xyz2 + (bb * n + k) * 3;
    const float dx = a[0] - o[0], dy = a[1] - o[1], dz = a[2] - o[2];
    dist[e] = (dx * dx + dy * dy) + dz * dz;
  }
}

__global__ __launch_bounds__(kThreads) void emd_bwd_kernel(
    int B, int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ graddist, const int *__restrict__ assignment,
    float *__restrict__ grad) {
  const long total = (long)B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long bb = e / n;
    const float *a = xyz1 + e * 3, *o = xyz2 + (bb * n + assignment[e]) * 3;
    const float g = graddist[e] * 2;
    grad[e * 3 + 0] = g * (a[0] - o[0]);
    grad[e * 3 + 1] = g * (a[1] - o[1]);
    grad[e * 3 + 2] = g * (a[2] - o[2]);
  }
}

EmdWs carve(void *workspace, int b, int n) {
  char *p = static_cast<char *>(workspace);
  const size_t arr = sn::align_up((size_t)b * n * 4, 256);
  EmdWs ws;
  ws.assignment_inv = reinterpret_cast<int *>(p); p += arr;
  ws.price = reinterpret_cast<float *>(p); p += arr;
  ws.bid = reinterpret_cast<int *>(p); p += arr;
  ws.bid_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_idx = reinterpret_cast<int *>(p); p += arr;
  ws.list[0] = reinterpret_cast<int *>(p); p += arr;
  ws.list[1] = reinterpret_cast<int *>(p); p += arr;
  ws.cnt[0] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * 4, 256);
  ws.cnt[1] = reinterpret_cast<int *>(p);
  return ws;
}

}  // namespace

extern "C" size_t sn_emd_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  return 8 * sn::align_up((size_t)b * n * 4, 256) + 2 * sn::align_up((size_t)b * 4, 256);
}

extern "C" int sn_emd_forward(const float *xyz1, const float *xyz2, int b, int n, float eps,
                              int iters, float *dist, int *assignment, void *workspace,
                              size_t workspace_bytes, long long *stats, void *stream) {
  SN_REQUIRE(xyz1 && xyz2 && dist && assignment && workspace, "sn_emd_forward: null pointer");
  SN_REQUIRE(b >= 1 && b <= 512, "sn_emd_forward: batch size must be in [1,512] (got %d)", b);
  SN_REQUIRE(n >= 1024 && n % 1024 == 0 && n <= (1 << 20),
             "sn_emd_forward: n must be a multiple of 1024, <= 2^20 (got %d)", n);
  SN_REQUIRE(iters >= 0, "sn_emd_forward: iters must be >= 0");
  SN_REQUIRE(workspace_bytes >= sn_emd_workspace_bytes(b, n),
             "sn_emd_forward: workspace too small (%zu < %zu)", workspace_bytes,
             sn_emd_workspace_bytes(b, n));
  hipStream_t s = sn::as_stream(stream);
  const EmdWs ws = carve(workspace, b, n);
  const long total = (long)b * n;
  const int eblocks = (int)((total + kThreads - 1) / kThreads < 2048 ? (total + kThreads - 1) / kThreads : 2048);
  emd_init_kernel<<<eblocks, kThreads, 0, s>>>(b, n, assignment, ws);
  const dim3 bid_grid(kBlocksPerCloud, b);
  const dim3 lin_grid(sn::ceil_div(n, kThreads * 4) < 16 ? sn::ceil_div(n, kThreads * 4) : 16, b);
  for (int it = 0; it < iters; ++it) {
    const int c = it & 1;
    SN_TIMED("emd_bid", s, (emd_bid_kernel<<<bid_grid, kThreads, 0, s>>>(
        n, xyz1, xyz2, eps, ws.price, ws.bid, ws.bid_inc, ws.max_inc, ws.max_idx, ws.list[c],
        ws.cnt[c], stats)));
    emd_getmax_kernel<<<lin_grid, kThreads, 0, s>>>(n, ws.bid, ws.bid_inc, ws.max_inc, ws.max_idx,
                                                    ws.list[c], ws.cnt[c], ws.cnt[c ^ 1]);
    emd_assign_kernel<<<lin_grid, kThreads, 0, s>>>(n, assignment, ws.assignment_inv, ws.price,
                                                    ws.bid, ws.bid_inc, ws.max_inc, ws.max_idx,
                                                    ws.list[c], ws.cnt[c], ws.list[c ^ 1],
                                                    ws.cnt[c ^ 1], it == iters - 1);
  }
  emd_calcdist_kernel<<<eblocks, kThreads, 0, s>>>(b, n, xyz1, xyz2, assignment, dist);
  return sn::launch_status("sn_emd_forward");
}

extern "C" int sn_emd_backward(const float *xyz1, const float *xyz2, const float *graddist,
                               const int *assignment, int b, int n, float *gradxyz1,
                               void *stream) {
  SN_REQUIRE(xyz1 && xyz2 && graddist && assignment && gradxyz1, "sn_emd_backward: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1, "sn_emd_backward: need b,n >= 1");
  const long total = (long)b * n;
  const int blocks = (int)((total + kThreads - 1) / kThreads < 2048 ? (total + kThreads - 1) / kThreads : 2048);
  emd_bwd_kernel<<<blocks, kThreads, 0, sn::as_stream(stream)>>>(b, n, xyz1, xyz2, graddist,
                                                                 assignment, gradxyz1);
  return sn::launch_status("sn_emd_backward");
}
