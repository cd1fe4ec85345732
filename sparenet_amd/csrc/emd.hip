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
//   reference's chunk of the 2048-tile (emd_cuda.cu:136-139) -- tracked on the exact
//   path through a canonical key, because the top-2 VALUES are partition independent;
//   GetMax window +-1e-6 in double; several bidders inside the window -> highest
//   bidder index (what a sequential ascending-j run of :188-191 gives; a race on
//   the reference's GPU); Assign/eviction/price update/last-iteration force
//   assignment exactly as :196-215.
//
// MI355X design (measured history in DESIGN.md section 5)
//   * ONE launch per call for all iterations (the reference: 7 launches per iteration): a persistent
//     kernel in which a team of workgroups owns a cloud and walks compact -> bid -> GetMax -> Assign with
//     team barriers in between (emd_auction_kernel below).
//   * Bid is the hot phase.  The pairwise search is a filtered one: a [targets x 4].[4 x bidders]
//     fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32) gives |t|^2 - 2 t.x for 256 pairs per
//     instruction and a per-lane threshold decides which pairs can still enter a bidder's
//     top-2; the rare survivors are queued in LDS and evaluated 64 at a time with the
//     reference's exact arithmetic (correctly rounded sqrt, fp64 detour).  Thresholds are seeded
//     from the bidder's previous two favourites (first iteration: two near targets found
//     through a Hilbert sort of the targets).  Results stay bit-identical.
//   * bidders are served in Hilbert order, 64 neighbours per group, S = 2^k <= 16 waves of a workgroup
//     sharing a group (each takes the superblocks sb with sb mod S == its segment); superblocks and
//     16-target blocks outside the reach of the group's filters are skipped by bounding box.
//   * GetMax: deterministic atomicMax of the bidder index inside the window.
#include <cstdlib>

#include "cloud_sort.hpp"
#include "common.hpp"

namespace {

constexpr int kThreads = 256;     // element-wise kernels
#ifndef SN_EMD_WAVES
#define SN_EMD_WAVES 4
#endif
constexpr int kRankBins = 256;    // per cloud: counters of unassigned bidders per 1/256 of the Hilbert ranks


struct Top2 {
  float best, better;
  int best_i, better_i;  // best_i: canonical among exact ties (see tie_key); better_i: a hint
};

__device__ __forceinline__ float bid_value(float tx, float ty, float tz, float p, float x1,
                                           float y1, float z1) {
#pragma clang fp contract(off)
  const float dx = tx - x1, dy = ty - y1, dz = tz - z1;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const float s = (xx + yy) + zz;
  return (float)((3.0 - (double)__builtin_sqrtf(s)) - (double)p);
}

__device__ __forceinline__ float sq_dist(float tx, float ty, float tz, float x1, float y1,
                                         float z1) {
#pragma clang fp contract(off)
  const float dx = tx - x1, dy = ty - y1, dz = tz - z1;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}

// Exact ties at the top.  The reference's Bid resolves d_k == best to
// argmin (thread(k), k): thread(k) = ((k mod 2048) / delta), delta = ceil(end_k / tpu)
// (emd_cuda.cu:136-139, :166-173).  key(k) = thread(k) * 2^20 + k orders those candidates.
struct TieGeom {
  int n, tpu;
};
__device__ __forceinline__ int tie_key(const TieGeom &g, int k) {
  const int k2 = (k / 2048) * 2048;
  const int end_k = (g.n < k2 + 2048 ? g.n : k2 + 2048) - k2;
  const int delta = (end_k + g.tpu - 1) / g.tpu;
  return ((k - k2) / delta) * (1 << 20) + k;  // n <= 2^20 (host check)
}

// if (d > best) {better = best; best = d; best_i = k} else if (d > better) better = d,
// plus: on d == best the candidate with the smaller tie key becomes best_i (values unchanged:
// better becomes best through the "else if").  Runs only on the exact path.
__device__ __forceinline__ void top2_push(Top2 &t, float d, int k, const TieGeom &g) {
  if (__any(d == t.best && t.best_i >= 0)) {  // rare: an exact tie with the running best
    if (d == t.best && t.best_i >= 0 && tie_key(g, k) < tie_key(g, t.best_i)) {
      const int o = t.best_i;
      t.best_i = k;
      k = o;  // the displaced index is an equally valid witness for `better`
    }
  }
  const bool gt = d > t.best;
  const bool mid = !gt && d > t.better;
  t.better_i = gt ? t.best_i : (mid ? k : t.better_i);
  t.better = gt ? t.best : (mid ? d : t.better);
  t.best_i = gt ? k : t.best_i;
  t.best = gt ? d : t.best;
}

// top-2 of the union of two partial results; equal best values keep the smaller tie key
__device__ __forceinline__ void top2_merge(Top2 &a, float b_best, float b_better, int b_i,
                                           int b_i2, const TieGeom &g) {
  if (b_best > a.best) {
    const bool keep_a = a.best >= b_better;
    a.better = keep_a ? a.best : b_better;
    a.better_i = keep_a ? a.best_i : b_i2;
    a.best = b_best;
    a.best_i = b_i;
  } else {
    if (b_best == a.best && b_i >= 0 && a.best_i >= 0 && tie_key(g, b_i) < tie_key(g, a.best_i)) {
      const int o = a.best_i;
      a.best_i = b_i;
      b_i = o;
    }
    const bool take_b = b_best > a.better;
    a.better = take_b ? b_best : a.better;
    a.better_i = take_b ? b_i : a.better_i;
  }
}

// ---- conservative fp32 filter --------------------------------------------------------
// A target k can change a lane's top-2 only if d_k > c, c = the lane's running `better`
// (or any proven lower bound of the bidder's final `better`).  With q = sqrtf(s):
//   d_k > c  =>  3 - q - p_k > c - 1e-15  =>  q < (3 - p_k - c) + 1e-15  =: R
//   =>  s < R^2 (1 + 2^-22).
// The filter evaluates R' = A'_k - c' in fp32 with A'_k = fl(3 - p_k) + eps (3 + |p_k|) and
// c' = c - eps (3 + |c|), eps = 2^-20: the two margins exceed every rounding error of the
// filter itself (<= 2^-22 (6 + |p| + |c|), plus 2^-22 relative on the FMA-evaluated s) and
// the relative slack needed on R, so
// "s <= R' |R'|" is implied by d_k >= c.  Only targets that pass go through the exact
// path (correctly rounded sqrt, fp64 detour, top-2 update); everything else costs
// 8 (distance) + 3 (filter) VALU ops instead of ~45.
constexpr float kFilterEps = 9.5367431640625e-07f;  // 2^-20

__device__ __forceinline__ float filter_target(float p) {
  return (3.0f - p) + (3.0f + __builtin_fabsf(p)) * kFilterEps;
}
__device__ __forceinline__ float filter_thr(float c) {
  return c - (3.0f + __builtin_fabsf(c)) * kFilterEps;
}
__device__ __forceinline__ bool filter_pass(float s, float a_k, float cthr) {
  const float r = a_k - cthr;
  return s <= r * __builtin_fabsf(r);
}

// order-preserving float max through integer atomics
__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
  if (v >= 0.f)
    __hip_atomic_fetch_max(reinterpret_cast<int *>(addr), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_fetch_min(reinterpret_cast<unsigned *>(addr), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- coherent accesses for data that workgroups hand to each other INSIDE the persistent launch.
// A relaxed agent-scope atomic load / store is a plain global_load / global_store with the sc1 bit: it
// bypasses the per-CU vector L1 (never refreshed by other CUs' stores) and is coherent across the XCDs'
// L2s, so the team barrier needs no release / acquire fence -- no L1 / L2 invalidation or write-back, the
// read-only streams (targets, MFMA operands, boxes, permutations) stay cached from iteration to iteration.
// Rule of the kernel: every word that is WRITTEN inside the launch is only ever read through these
// (or through atomics) and written through stc below; words written by earlier launches only are read with
// plain loads.  One deliberate exception: `prt` (see bid_group), where a stale value is a valid bound.
__device__ __forceinline__ int ldc(const int *p) {
  return (int)__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ldc(const float *p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// Stores.  An sc1 (agent-scope) store of 4 bytes is one fabric write and DROPS the line from the XCD's L2, so
// the next coherent load of it -- even from the same XCD -- is served over the fabric.  When every workgroup of
// a team sits on ONE XCD (`loc`, established at team formation in the kernel), a plain store is enough: the L1
// is write-through, the team's coherent loads bypass their L1s and meet in that XCD's L2, where the line now
// stays.  Measured at B = 32: 2.95 -> 2.78 ms per call, the iterations with many stores gain most.
__device__ __forceinline__ void stc(bool loc, int *p, int v) {
  if (loc)
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc(bool loc, float *p, float v) {
  stc(loc, reinterpret_cast<int *>(p), __float_as_int(v));
}
// {price, target index} of a stream position: ONE coherent 8-byte load (the price changes inside the launch)
__device__ __forceinline__ float2 ldc_pk(const float2 *p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ int2 ldc2(const int *p) {  // 8-byte aligned pair
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
}

// Prepared target data per cloud, all in Morton order (stream position p holds target tperm[p]):
//  * t4s[p] = {x, y, z, k} (constant) and pk[p] = {price_k, k}: what the precise filter and the exact path
//    read, both addressed by the stream position of a hit (one round trip).  Written by emd_init_kernel,
//    the price refreshed by the Assign phase for the targets whose price changed (rank2[k] = p); the same
//    price goes to prt, the transposed copy the coarse filter reads.
//  * mstream: the MFMA A-operand of the coarse filter, price independent (written once).
//    u_kj = |t_k|^2 - 2 t_k . x_j is a [targets x 4] . [4 x bidders] product with rows
//    (-2x, -2y, -2z, |t|^2) and columns (x, y, z, 1).  v_mfma_f32_16x16x4_f32 takes ONE float
//    per lane for A: lane l supplies A[i = l & 15][k = l >> 4].  A superblock of 64 targets is
//    4 such operands; lane l's four values sit in one float4:
//      mstream[(superblock * 64 + l)].q = component (l >> 4) of target 64 sb + 16 q + (l & 15)
//    so a wave fetches 64 targets with one coalesced global_load_dwordx4 per lane.
//  * sbbox: the bounding box of every block of 16 targets, for the pruning test.
typedef float f4 __attribute__((ext_vector_type(4)));

struct EmdWs {
  int *assignment_inv;
  float *price;
  int *bid, *bid2;
  float *bid_inc;
  float *max_inc;
  int *max_idx;  // GetMax's winner per target; PERSISTS across iterations like the reference's tensor
  int *win;      // this iteration's in-window winner per target (-1: nobody was in the window)
  int *list[1];  // the unassigned bidders of the iteration, per workgroup in its own rank range
  float *prt;    // [B, n/64, 16, 4] prices by stream position, transposed for the coarse filter (see bid_group)
  int *bins[2];  // [B, 64] ping-pong: flagged (= unassigned) bidders per 1/64 of the rank range
  f4 *t4s;       // [B, n] by stream position p: {x, y, z, index bits} of target tperm[p] (constant in the launch)
  float2 *pk;    // [B, n] by stream position: {price, target index bits}
  int *rank2;    // [B, n] target index -> stream position
  f4 *mstream;   // [B, n/64, 64], targets in Morton order (position p holds target tperm[p])
  int *tperm;    // [B, n] sorted position -> target index
  int *cell_of;  // [B, n] sort scratch
  int *hist;     // [B, 4096] cell offsets of the sorted targets
  float *bbox;   // [B, 6] bounding box of the targets (also bounds |t|^2 for the filter slack)
  float *sbbox;  // [B, n/16, 8] bounding box of every block of 16 targets of the stream
  int *perm1;    // [B, n] Morton rank -> bidder index
  int *rank1;    // [B, n] bidder index -> Morton rank
  int *flags;    // [B, n] by rank: unassigned after this iteration (next list = flagged ranks in order)
  int *hist1;    // [B, 4096] sort scratch of the bidders
  float *bbox1;  // [B, 6]
  void *ctl;     // persistent auction: ticket, abort word, one barrier counter per team
};

__global__ void emd_init_kernel(int B, int n, const float *__restrict__ xyz2,
                                int *__restrict__ assignment, EmdWs ws) {
#pragma clang fp contract(off)
  const long total = (long)B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    assignment[e] = -1;
    ws.assignment_inv[e] = -1;
    ws.price[e] = 0.f;
    ws.max_inc[e] = 0.f;  // emd_module.py:49 (zeros, not -1e9)
    ws.max_idx[e] = 0;
    ws.bid[e] = -1;   // no previous favourites yet (filter seeding)
    ws.bid2[e] = -1;
    ws.list[0][e] = ws.perm1[e];  // Morton order: 64 consecutive bidders are neighbours
    ws.rank1[e - e % n + ws.perm1[e]] = (int)(e % n);
    ws.flags[e] = 1;  // every bidder starts flagged (= unassigned)
    ws.prt[e] = 0.f;
    {  // stream position p of this cloud holds target k = tperm[p]
      const long bb = e / n;
      const int p = (int)(e - bb * n);
      const int k = ws.tperm[e];
      const float *t = xyz2 + (bb * n + k) * 3;
      const float x = t[0], y = t[1], z = t[2];
      ws.t4s[e] = f4{x, y, z, __int_as_float(k)};
      ws.pk[e] = make_float2(0.f, __int_as_float(k));
      ws.rank2[bb * n + k] = p;
      const float tt = (x * x + y * y) + z * z;
      float *m = reinterpret_cast<float *>(ws.mstream + (bb * (n >> 6) + (p >> 6)) * 64);
      const int q = (p >> 4) & 3, c = p & 15;
      m[(0 * 16 + c) * 4 + q] = -2.f * x;
      m[(1 * 16 + c) * 4 + q] = -2.f * y;
      m[(2 * 16 + c) * 4 + q] = -2.f * z;
      m[(3 * 16 + c) * 4 + q] = tt;
    }
    if (e < (long)B * kRankBins) {
      ws.bins[0][e] = n / kRankBins;  // every bidder starts unassigned
      ws.bins[1][e] = 0;
    }
  }
}

// min / max over each row of 16 lanes, left in every lane of the row (four DPP steps)
__device__ __forceinline__ float row16_min(float v) {
  v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)));
  v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)));
  v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true)));
  v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true)));
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)));
  v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)));
  v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true)));
  v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true)));
  return v;
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// bounding box of every block of 16 consecutive targets of the Morton-ordered stream (the rows of one
// MFMA A operand); four of them make a superblock
__global__ __launch_bounds__(256) void emd_sbbox_kernel(int B, int n, const float *__restrict__ xyz2,
                                                        EmdWs ws) {
  const long sb_all = (long)B * (n >> 6);
  const int lane = threadIdx.x & 63;
  for (long sb = (long)blockIdx.x * 4 + (threadIdx.x >> 6); sb < sb_all; sb += (long)gridDim.x * 4) {
    const long bb = sb / (n >> 6);
    const float *t = xyz2 + (bb * n + ws.tperm[sb * 64 + lane]) * 3;
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = row16_min(t[a]);
      hi[a] = row16_max(t[a]);
    }
    const int c = lane & 15;
    if (c < 8)
      ws.sbbox[(sb * 4 + (lane >> 4)) * 8 + c] = c == 0 ? lo[0] : c == 1 ? lo[1] : c == 2 ? lo[2]
                                               : c == 3 ? hi[0] : c == 4 ? hi[1] : c == 5 ? hi[2] : 0.f;
  }
}

// First-iteration seeds.  The bid filter needs, per bidder, two real targets whose values
// bound the final `better` from below; later iterations use the previous favourites, the first
// one has none and would evaluate ~75 targets per bidder exactly before its thresholds
// tighten.  Targets are already in Morton order: a window of 16 sorted positions around the
// bidder's own cell supplies candidates, the two nearest become bid/bid2.  ANY two distinct
// targets are valid seeds; better ones only make the filter reject more.
__global__ __launch_bounds__(kThreads) void emd_seed_kernel(int B, int n,
                                                            const float *__restrict__ xyz1,
                                                            const float *__restrict__ xyz2,
                                                            EmdWs ws) {
#pragma clang fp contract(off)
  const long total = (long)B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    // threads walk the bidders in sorted (Hilbert) order: neighbouring threads then read overlapping windows
    // of the target stream out of the L1 instead of 384 scattered bytes each
    const long bb = e / n;
    const long je = bb * n + ws.perm1[e];
    const float x = xyz1[je * 3 + 0], y = xyz1[je * 3 + 1], z = xyz1[je * 3 + 2];
    const float *box = ws.bbox + bb * 6;
    unsigned q[3];
    const float v[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float ext = box[3 + a] - box[a];
      q[a] = sort_coord(v[a], box[a], sort_scale(ext));
    }
    const int c = (int)morton3_4bit(q[0], q[1], q[2]);
    const int start = c > 0 ? ws.hist[bb * kSortCells + c - 1] : 0;  // END of the previous cell
    int lo = start - 4;
    lo = lo < 0 ? 0 : (lo > n - 16 ? n - 16 : lo);
    float s1 = 3e38f, s2 = 3e38f;
    int k1 = -1, k2 = -1;
    for (int p = lo; p < lo + 16; ++p) {  // the prepared stream: one 16-byte record per position, no second gather
      const f4 t = ws.t4s[bb * n + p];
      const int k = __float_as_int(t.w);
      const float dx = t.x - x, dy = t.y - y, dz = t.z - z;
      const float sq = (dx * dx + dy * dy) + dz * dz;
      if (sq < s1) {
        s2 = s1;
        k2 = k1;
        s1 = sq;
        k1 = k;
      } else if (sq < s2) {
        s2 = sq;
        k2 = k;
      }
    }
    ws.bid[je] = k1;
    ws.bid2[je] = k2;
  }
}

struct BidOut {
  int *bid, *bid2;
  float *bid_inc, *max_inc;
  int *win;
  bool loc;  // the team sits on one XCD: plain stores (see stc)
};

constexpr int kBidWaves = 16;
constexpr int kBidThreads = kBidWaves * 64;

__device__ __forceinline__ void emit_bid(const BidOut &A, size_t o, int j, const Top2 &top,
                                         float eps) {
  const bool loc = A.loc;
  if (top.best_i < 0) {  // only with non-finite coordinates: no comparison succeeded
    stc(loc, &A.bid[o + j], -1);
    stc(loc, &A.bid2[o + j], -1);
    stc(loc, &A.bid_inc[o + j], 0.f);
    return;
  }
  const float inc = (top.best - top.better) + eps;
  stc(loc, &A.bid[o + j], top.best_i);
  stc(loc, &A.bid2[o + j], top.better_i == top.best_i ? -1 : top.better_i);
  stc(loc, &A.bid_inc[o + j], inc);
  atomic_max_float(&A.max_inc[o + top.best_i], inc);
  stc(loc, &A.win[o + top.best_i], -1);  // this iteration's winner is derived in the GetMax phase
}

// ---------------------------------------------------------------------------------------
// Bid phase: a two-level filter in front of the exact evaluation.
//
// level 1 (matrix cores).  A wave serves 64 bidders and walks its target segment in
//   superblocks of 64 targets.  For bidder group g (16 bidders) and target block q (16 targets)
//   one v_mfma_f32_16x16x4_f32 returns u = |t|^2 - 2 t.x for the 256 pairs; lane l receives
//   the four targets 16 q + 4 (l >> 4) + r of bidder 16 g + (l & 15).  A pair can only matter
//   if u <= T'_j, T'_j = Rmax |Rmax| (1 + 2^-20) - |x_j|^2 + slack, Rmax = A'max - c'_j: the
//   per-target A'_k of the precise filter is replaced by its upper bound over all targets
//   (prices never fall below `price_floor`), which makes the threshold a per-LANE constant.
//   Cost: 16 MFMA (32 cycles each, exact fp32 = an fmaf chain) + ~44 VALU per 4096 pairs.
//   The slack 2^-18 (max|t|^2 + |x|^2) covers the fmaf chain's rounding (4 roundings of
//   partial sums <= 2 (|t|^2 + |x|^2)), the rounding of the stored |t|^2 and of |x|^2, and
//   the fp32 evaluation of T' itself.
// hit queue.  Level-1 hits are rare and scattered over the lanes, so they are not evaluated
//   in place: (target, bidder) pairs are appended to a per-wave LDS queue and handled 64 at a
//   time with every lane busy.
// level 2 (precise filter) + exact path, per queued pair.  t4s / pk of the position are loaded and the
//   exact-arithmetic test s <= R' |R'|, R' = A'_k - c'_j (derivation above) applied; the
//   survivors get the reference's arithmetic (correctly rounded sqrt, fp64 detour) and are
//   pushed into the bidder's top-2, which lives in LDS; lanes holding pairs of the same bidder
//   take turns (an election through LDS per round).
//   d_k >= c  =>  level 2 passes  =>  level 1 passes, and a stale (smaller) c only lets more
//   through, so the top-2 VALUES are those of the full scan; exact ties resolve through
//   tie_key, which does not depend on the visiting order either.
//
// superblock pruning.  Bidders are served in Morton order (the unassigned list is rebuilt in
//   rank order every iteration), so the 64 bidders of a wave are neighbours; the wave keeps
//   their bounding box and the largest reach of their filters and only visits the superblocks
//   whose bounding box lies within that reach (64 box tests at a time, lane = superblock).
// Work split: S = 2^k <= 16 waves of ONE workgroup share a group of 64 bidders, wave s taking
// the superblocks sb with sb mod S == s; their partial top-2's meet in LDS in arrival order.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float min16(const f4 a, const f4 b, const f4 c, const f4 d) {
  const float m0 = __builtin_fminf(__builtin_fminf(a.x, a.y), a.z);
  const float m1 = __builtin_fminf(__builtin_fminf(a.w, b.x), b.y);
  const float m2 = __builtin_fminf(__builtin_fminf(b.z, b.w), c.x);
  const float m3 = __builtin_fminf(__builtin_fminf(c.y, c.z), c.w);
  const float m4 = __builtin_fminf(__builtin_fminf(d.x, d.y), d.z);
  const float m5 = __builtin_fminf(__builtin_fminf(m0, m1), d.w);
  return __builtin_fminf(__builtin_fminf(m2, m3), __builtin_fminf(m4, m5));
}

__device__ __forceinline__ unsigned hits4(const f4 d, float thr, int shift) {
  return ((d.x <= thr ? 1u : 0u) | (d.y <= thr ? 2u : 0u) | (d.z <= thr ? 4u : 0u) |
          (d.w <= thr ? 8u : 0u)) << shift;
}

constexpr int kQueue = 128;  // a round appends <= 64 pairs to < 64 left-overs

struct WaveTab {  // per-wave LDS: the 64 bidders it serves, and its hit queue
  float x[64], y[64], z[64];
  float cm[64];  // proven lower bound of the bidder's final `better` (3e38: no bidder)
  float best[64], better[64];
  int bi[64], bi2[64];
  unsigned queue[kQueue];  // stream position | bidder << 20
  int owner[64];
};

struct GroupAcc {  // per bidder group of a workgroup: the arrival-order merge of its segments
  float best[64], better[64];
  int bi[64], bi2[64];
  int lock, arrived;
  // the group's bidders, looked up ONCE by the group's first wave (index, coordinates, the threshold seed from the
  // two previous favourites: four dependent gathers).  Sixteen waves each doing these lookups for the same 64
  // bidders put 16 x 12 x 64 scattered requests on the CU's address unit per iteration: 3.1 us of every bid phase.
  float sx[64], sy[64], sz[64], scm[64];
  int sj[64];
};

// level-1 threshold T' of one bidder: base = slack - |x|^2 is fixed, cm grows
__device__ __forceinline__ float coarse_threshold(float cm, float base, float a_max) {
  const float r = a_max - filter_thr(cm);
  return __builtin_fmaf(r * __builtin_fabsf(r), 1.00000095367431640625f, base);
}

// =======================================================================================
// Persistent auction: ALL iterations of a call in ONE launch.
//
// A launch-per-phase form (bid / GetMax / Assign / compact kernels, 200 launches per call; round 1) pays
// four kernel boundaries and four cold grids per iteration.  Here a TEAM of G workgroups owns a cloud for
// the whole call and walks  compact -> bid -> [barrier] -> getmax -> [barrier] -> assign -> [barrier]  with a team barrier
// (monotonic counter; the exchanged words are read and written with coherent accesses: placement
// independent) between the phases.
//   * Workgroup m of a team owns the Morton RANKS [m n/G, (m+1) n/G) of the bidders for the whole call:
//     it compacts its own raised flags into a local list (no global scan), bids for those bidders, runs
//     their GetMax / Assign steps.  Its bidders stay spatial neighbours, and the targets near them stay in
//     its L1 / the team's L2 from one iteration to the next.  The order in which bidders are served never
//     enters a result (top-2 values, tie keys, atomicMax winners), so this is bit-identical.
//   * S = 2^k <= 16 waves share one local group of 64 bidders exactly as above.
//   * The unassigned count of the next iteration (the reference's tie geometry needs it) is accumulated
//     with one atomic per wave while Assign raises the flags.
//   * Teams are formed from a TICKET taken at start, not from blockIdx: a workgroup only ever waits for
//     workgroups that have started, and at most one block of teams per launch is incomplete at any time,
//     so concurrent launches on other streams cannot starve each other (each can hold <= 63 CUs waiting).
//     With >= 32 clouds a team is G consecutive tickets of ONE XCD's counter (the XCD read from the hardware
//     register: its L2 then keeps the cloud's streams; speed only); fewer clouds get larger, contiguous
//     teams from one global counter.
//   * Every spin is bounded; a timeout raises ctl.abort, every workgroup of the launch leaves, and
//     sn_emd_forward reports it on the next call that checks (SN_EMD_CHECK=1: immediately).
// =======================================================================================
struct AuctionCtl {  // zeroed by a memset node before every launch
  unsigned ticket;
  unsigned abort;
  unsigned xticket[8];  // one ticket counter per XCD (teams of the XCD-local geometry)
  unsigned pad[22];
  unsigned bar[1];  // [teams * 32]: one counter per team, 128 bytes apart
};

struct TeamGeom {
  int G;       // workgroups per team (power of two)
  int teams;   // teams in the launch
  int xcd;     // 1: a team = G consecutive tickets of one XCD's counter (XCD-local teams)
};

__host__ __device__ inline TeamGeom team_geometry(int B, int W) {
  TeamGeom t;
  if (B >= 32 && W >= 64) {
    // one XCD's share of a block of 64 tickets per team; more clouds than teams: halve the teams' size
    // until every cloud has its own team (teams of one workgroup serve several clouds in turn)
    int g = 8;
    while (g > 1 && (W / (8 * g)) * 8 < B) g >>= 1;
    t.G = g;
    t.teams = (W / (8 * g)) * 8;
    t.xcd = 1;
  } else {
    int g = 1;
    while (g < 64 && g * 2 * B <= W) g *= 2;
    t.G = g;
    t.teams = W / g > 0 ? W / g : 1;
    t.xcd = 0;
  }
  return t;
}

constexpr unsigned kSpinLimit = 40u * 1000u * 1000u;  // x ~64 ns: > 2 s

struct TeamSync {
  unsigned *bar;     // the team's counter
  unsigned *abort;   // the launch's abort word
  unsigned target;   // arrivals expected at the next barrier
  int G;
};

// All waves of all G workgroups arrive.  No fence: every word a phase hands to the next one is written and
// read through the coherent accessors above (ldc / stc / atomics); each wave drains its stores
// (s_waitcnt vmcnt(0)) before its workgroup arrives, the arrival counter is a device-scope atomic.
// Measured: 1.1 us per barrier instead of 3.7 us with an agent-scope release + acquire pair, and the phases
// after it no longer start with an invalidated L1 / L2.
__device__ __forceinline__ bool team_barrier(TeamSync &ts, int *s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ts.target += (unsigned)ts.G;
  if (threadIdx.x == 0) {
    int ok = 1;
    __hip_atomic_fetch_add(ts.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ts.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ts.target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u) {
        if (__hip_atomic_load(ts.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if (spins > kSpinLimit) {
          __hip_atomic_store(ts.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

struct BidCtx {
  int n, nsb;
  float eps, a_max, tmax;
  TieGeom geom;
  size_t o;
  const float *p1;   // bidders of this cloud
  const f4 *t4;      // by stream position; prices inside change between the phases (no __restrict__)
  const float2 *pkc;
  const int *rk2;
  const f4 *ms;      // MFMA operand stream of this cloud
  const f4 *prt;     // prices of this cloud, transposed (plain loads on purpose: see below); nullptr: not used
  const float *sbb;  // block boxes of this cloud
  BidOut A;
#ifdef SN_BID_STAMPS
  long long *stamps;  // experiment build: per-wave time per part of bid_group (100 MHz ticks), or nullptr
#endif
};

// One group of 64 bidders, seen by one of its S segment-waves.
__device__ __forceinline__ void bid_group(const BidCtx &c, WaveTab &T, GroupAcc &ga, const int *lst,
                                          int count, int grp, int ngroups, int S, int seg, int lane) {
  const int row = lane >> 4, col = lane & 15;
  const int u = grp * 64 + lane;
  const bool active = grp < ngroups && u < count;
  const float a_max = c.a_max, tmax = c.tmax;
  const TieGeom geom = c.geom;
  const f4 *t4 = c.t4;
  const float2 *pkc = c.pkc;
  Top2 top = {-1e9f, -1e9f, -1, -1};
  int j = 0;
#ifdef SN_BID_STAMPS
  long long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tk = (long long)__builtin_amdgcn_s_memrealtime();
#define STAMP(i) { const long long now_ = (long long)__builtin_amdgcn_s_memrealtime(); st[i] += now_ - tk; tk = now_; }
#define COUNT(i) st[i] += 1;
#else
#define STAMP(i)
#define COUNT(i)
#endif
  if (grp < ngroups && seg == 0) {  // wave-uniform: the group's first wave looks its bidders up
    const int jj = ldc(&lst[active ? u : grp * 64]);
    const float x1 = c.p1[jj * 3 + 0], y1 = c.p1[jj * 3 + 1], z1 = c.p1[jj * 3 + 2];
    float cm = -1e9f;
    const int pa = ldc(&c.A.bid[c.o + jj]), pb = ldc(&c.A.bid2[c.o + jj]);
    if (pa >= 0 && pb >= 0) {
      const int qa = c.rk2[pa], qb = c.rk2[pb];
      const f4 ta = t4[qa], tb = t4[qb];   // coordinates: constant; the price comes from pk
      const float da = bid_value(ta.x, ta.y, ta.z, ldc_pk(pkc + qa).x, x1, y1, z1);
      const float db = bid_value(tb.x, tb.y, tb.z, ldc_pk(pkc + qb).x, x1, y1, z1);
      cm = __builtin_fminf(da, db);
    }
    ga.sx[lane] = x1;
    ga.sy[lane] = y1;
    ga.sz[lane] = z1;
    ga.scm[lane] = cm;
    ga.sj[lane] = jj;
  }
  __syncthreads();  // every wave of the workgroup calls bid_group the same number of times
  if (grp < ngroups) {  // wave-uniform
    j = ga.sj[lane];
    float blo[4][3], bhi[4][3];
    float own_slack2;
    {
      const float x1 = ga.sx[lane], y1 = ga.sy[lane], z1 = ga.sz[lane];
      {
#pragma clang fp contract(off)
        const float xx = (x1 * x1 + y1 * y1) + z1 * z1;
        own_slack2 = 2.f * 3.814697265625e-06f * (tmax + xx);
        const float v[3] = {x1, y1, z1};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float lo = row16_min(active ? v[a] : 3.0e38f), hi = row16_max(active ? v[a] : -3.0e38f);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            blo[g][a] = lane_value(lo, 16 * g);
            bhi[g][a] = lane_value(hi, 16 * g);
          }
        }
      }
      const float cm = ga.scm[lane];
      T.x[lane] = x1;
      T.y[lane] = y1;
      T.z[lane] = z1;
      T.cm[lane] = active ? cm : 3.0e38f;
      T.best[lane] = -1e9f;
      T.better[lane] = -1e9f;
      T.bi[lane] = -1;
      T.bi2[lane] = -1;
    }
    // Level 1 with the target's own price (c.prt != nullptr: eps >= 0 and not the first iteration).  With
    // a_k = A'_k - 3 and gamma_j = c'_j - 3 (both small: the squares below do not cancel),
    //   s <= (A'_k - c'_j)^2   <=>   u_kj - a_k^2 + 2 a_k gamma_j  <=  gamma_j^2 - |x_j|^2,
    // and the left side is the first MFMA's result carried through a SECOND one with rows (-a_k^2, 2 a_k, 0, 0)
    // and columns (1, gamma_j, 0, 0).  The rows come from `prt`, the prices by stream position written by Assign
    // and read here with PLAIN loads: a stale line holds an earlier -- lower -- price of the same target
    // (prices only rise for eps >= 0), i.e. a larger A'_k, which only lets more pairs through.  The squared form
    // forgets the sign of A'_k - c'_j: a negative one passes spuriously and is rejected at level 2; a bidder
    // for whom even A'max - c'_j is negative gets the threshold -3e38.  Slack: eight fmaf roundings of partial
    // sums <= 2 (|t|^2 + |x|^2) + (|a| + |gamma|)^2 with |a| <= |gamma| for every pair level 2 can pass, the
    // rounding of a_k, a_k^2, gamma_j^2 and of level 2's own r |r|: 2^-17 (max|t|^2 + |x|^2) + 2^-16 gamma^2
    // covers them four times over.
    const bool up = c.prt != nullptr;  // wave-uniform
    float thr[4], bop[4], bop2[4];
    auto set_thresholds = [&]() {  // at the start and after every drain (rare with prices in level 1)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma clang fp contract(off)
        const int cc = 16 * g + col;
        const float x = T.x[cc], y = T.y[cc], z = T.z[cc];
        const float xx = (x * x + y * y) + z * z;
        const float base = 7.62939453125e-06f * (tmax + xx) - xx;  // 2^-17: see above (2^-18 would do without prices)
        const float cmv = T.cm[cc];
        if (!up) {
          thr[g] = coarse_threshold(cmv, base, a_max);
          bop2[g] = 0.f;
        } else {
          const float ct = filter_thr(cmv);
          const bool open = a_max - ct >= 0.f;  // false for idle lanes (cm = 3e38) as well
          const float gam = ct - 3.0f;
          thr[g] = open ? __builtin_fmaf(gam * gam, 1.0000152587890625f, base) : -3.0e38f;
          bop2[g] = row == 0 ? 1.0f : (row == 1 && open ? gam : 0.f);
        }
      }
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cc = 16 * g + col;
      bop[g] = row == 0 ? T.x[cc] : (row == 1 ? T.y[cc] : (row == 2 ? T.z[cc] : 1.0f));
    }
    set_thresholds();
    int qcount = 0;

    auto batch = [&](int first, int cnt) {
      const bool on = lane < cnt;
      const unsigned e = T.queue[first + (on ? lane : 0)];
      const int cc = (int)(e >> 20);
      const f4 t = t4[e & 0xfffffu];              // x, y, z (constant inside the launch)
      const float2 pq = ldc_pk(pkc + (e & 0xfffffu));  // today's price + the target's index
      const int k = __float_as_int(pq.y);
      const float sq = sq_dist(t.x, t.y, t.z, T.x[cc], T.y[cc], T.z[cc]);
      bool pend = on && filter_pass(sq, filter_target(pq.x), filter_thr(T.cm[cc]));
      float d = 0.f;
      if (pend) d = (float)((3.0 - (double)__builtin_sqrtf(sq)) - (double)pq.x);
#ifdef SN_BID_STAMPS
      st[12] += __popcll(__ballot(pend));
#endif
      volatile int *own = T.owner;
      while (__any(pend)) {
        COUNT(13)
        asm volatile("" ::: "memory");
        if (pend) own[cc] = lane;
        if (pend && own[cc] == lane) {
          Top2 tp = {T.best[cc], T.better[cc], T.bi[cc], T.bi2[cc]};
          top2_push(tp, d, k, geom);
          T.best[cc] = tp.best;
          T.better[cc] = tp.better;
          T.bi[cc] = tp.best_i;
          T.bi2[cc] = tp.better_i;
          T.cm[cc] = __builtin_fmaxf(T.cm[cc], tp.better);
          pend = false;
        }
      }
      asm volatile("" ::: "memory");
    };

    float r2g[4];
    auto refresh_reach = [&]() {
      const float v = row16_max(active ? coarse_threshold(T.cm[lane], own_slack2, a_max) : -3.0e38f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float r = lane_value(v, 16 * g);
        r2g[g] = r > 0.f ? r * 1.0001f : r;
      }
    };
    refresh_reach();
    STAMP(0)
    const f4 *ms = c.ms + lane;
    const f4 *prt = c.prt + col;
    const float *sbb = c.sbb;
    auto worth = [&](const f4 lo4, const f4 hi4) {
      unsigned m = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float gx = __builtin_fmaxf(__builtin_fmaxf(lo4.x - bhi[g][0], blo[g][0] - lo4.w), 0.f);
        const float gy = __builtin_fmaxf(__builtin_fmaxf(lo4.y - bhi[g][1], blo[g][1] - hi4.x), 0.f);
        const float gz = __builtin_fmaxf(__builtin_fmaxf(lo4.z - bhi[g][2], blo[g][2] - hi4.y), 0.f);
        m |= (((gx * gx + gy * gy) + gz * gz) * 0.9999f <= r2g[g] ? 1u : 0u) << g;
      }
      return m;
    };
    auto quad_mask = [&](unsigned m) {
      const unsigned m0 = (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0x00, 0xf, 0xf, true);
      const unsigned m1 = (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0x55, 0xf, 0xf, true);
      const unsigned m2 = (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0xAA, 0xf, 0xf, true);
      const unsigned m3 = (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0xFF, 0xf, 0xf, true);
      return m0 | (m1 << 4) | (m2 << 8) | (m3 << 12);
    };
    const int owned4 = (c.nsb / S) * 4;
    for (int t0 = 0; t0 < owned4; t0 += 64) {
      const int task = t0 + lane;
      const int sbl = (task >> 2) * S + seg;
      const bool mine = task < owned4;
      f4 box_lo = {0.f, 0.f, 0.f, 0.f}, box_hi = {0.f, 0.f, 0.f, 0.f};
      if (mine) {
        box_lo = *reinterpret_cast<const f4 *>(sbb + (size_t)(sbl * 4 + (task & 3)) * 8);
        box_hi = *reinterpret_cast<const f4 *>(sbb + (size_t)(sbl * 4 + (task & 3)) * 8 + 4);
      }
      unsigned gmask = quad_mask(mine ? worth(box_lo, box_hi) : 0u);
      unsigned long long todo = __ballot(gmask != 0u && (lane & 3) == 0);
      // The operand of a visit is ALWAYS the one requested during the previous visit, and every visit requests
      // exactly one (the last one asks for its own again): with a conditional request hipcc has to wait for
      // "all loads" in front of the MFMAs, which serialised every visit behind the next operand's L2 latency.
      f4 a_next = {0.f, 0.f, 0.f, 0.f}, p_next = {0.f, 0.f, 0.f, 0.f};
      int next_sb = 0;
      if (todo) {
        next_sb = ((t0 + __builtin_ctzll(todo)) >> 2) * S + seg;
        a_next = ms[(size_t)next_sb * 64];
        if (up) p_next = prt[(size_t)next_sb * 16];
      }
      STAMP(1)
      while (todo) {
        COUNT(6)
        const int tl = __builtin_ctzll(todo);
        const int sb = ((t0 + tl) >> 2) * S + seg;
        todo &= todo - 1;
        const int kb = sb * 64;
        const f4 a = a_next, pr = p_next;
        next_sb = todo ? ((t0 + __builtin_ctzll(todo)) >> 2) * S + seg : sb;
        a_next = ms[(size_t)next_sb * 64];
        f4 a2 = {0.f, 0.f, 0.f, 0.f};
        if (up) {
          p_next = prt[(size_t)next_sb * 16];
          const float t0_ = filter_target(pr.x) - 3.0f, t1_ = filter_target(pr.y) - 3.0f;
          const float t2_ = filter_target(pr.z) - 3.0f, t3_ = filter_target(pr.w) - 3.0f;
          a2.x = row == 0 ? -(t0_ * t0_) : (row == 1 ? 2.f * t0_ : 0.f);
          a2.y = row == 0 ? -(t1_ * t1_) : (row == 1 ? 2.f * t1_ : 0.f);
          a2.z = row == 0 ? -(t2_ * t2_) : (row == 1 ? 2.f * t2_ : 0.f);
          a2.w = row == 0 ? -(t3_ * t3_) : (row == 1 ? 2.f * t3_ : 0.f);
        }
#ifdef SN_BID_STAMPS
        STAMP(2)
        if (todo) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(5)
#endif
        bool drained = false;
        const unsigned gm = (unsigned)__builtin_amdgcn_readlane((int)gmask, tl);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (!(gm & (0x1111u << g))) continue;
          COUNT(8)
          // all four blocks of the superblock once the subgroup reaches any of them: a skipped MFMA costs
          // a branch and four moves of "far" into its result registers on the vector ALU, which is the busy
          // unit here -- the matrix pipe is not (a block out of reach cannot produce a hit that matters)
          const f4 zero = {0.f, 0.f, 0.f, 0.f};
          f4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bop[g], zero, 0, 0, 0);
          f4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bop[g], zero, 0, 0, 0);
          f4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bop[g], zero, 0, 0, 0);
          f4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bop[g], zero, 0, 0, 0);
          if (up) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, bop2[g], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, bop2[g], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, bop2[g], d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, bop2[g], d3, 0, 0, 0);
          }
          if (__builtin_expect(__any(min16(d0, d1, d2, d3) <= thr[g]), 0)) {
            unsigned hm = hits4(d0, thr[g], 0) | hits4(d1, thr[g], 4) | hits4(d2, thr[g], 8) |
                          hits4(d3, thr[g], 12);
            COUNT(9)
            STAMP(2)
            while (__any(hm != 0)) {
              COUNT(10)
              const bool has = hm != 0;
              const int i = has ? __builtin_ctz(hm) : 0;
              hm &= hm - 1;
              const unsigned long long bal = __ballot(has);
              const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi(
                                           (unsigned)(bal >> 32),
                                           __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
              if (has)
                T.queue[pos] = (unsigned)(kb + 16 * (i >> 2) + 4 * row + (i & 3)) |
                               ((unsigned)(16 * g + col) << 20);
              qcount += __popcll(bal);
#ifdef SN_BID_STAMPS
              st[11] += __popcll(bal);
#endif
              while (qcount >= 64) {
                qcount -= 64;
                STAMP(14)
                batch(qcount, 64);
                STAMP(3)
                COUNT(7)
                drained = true;
              }
            }
            STAMP(14)
          }
        }
        if (drained) {
          set_thresholds();
          refresh_reach();
          if (todo) {  // the lane's block box again (not kept in registers across the visits: drains are rare)
            const bool left = (todo >> (lane & ~3)) & 1ull;
            f4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
            if (left) {
              lo4 = *reinterpret_cast<const f4 *>(sbb + (size_t)(sbl * 4 + (task & 3)) * 8);
              hi4 = *reinterpret_cast<const f4 *>(sbb + (size_t)(sbl * 4 + (task & 3)) * 8 + 4);
            }
            gmask = quad_mask(left ? worth(lo4, hi4) : 0u);
            todo = __ballot(gmask != 0u && (lane & 3) == 0);
            if (todo) {  // the tightened reach may have dropped the superblock whose operand is on its way
              const int nsb = ((t0 + __builtin_ctzll(todo)) >> 2) * S + seg;
              if (nsb != next_sb) {
                next_sb = nsb;
                a_next = ms[(size_t)next_sb * 64];
                if (up) p_next = prt[(size_t)next_sb * 16];
              }
            }
          }
        }
        STAMP(2)
      }
    }
    if (qcount > 0) batch(0, qcount);
    STAMP(4)
    top = Top2{T.best[lane], T.better[lane], T.bi[lane], T.bi2[lane]};
  }
  bool emit = seg == 0;
  if (S > 1 && grp < ngroups) {
    if (lane == 0)
      while (atomicCAS(&ga.lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int arrived = ga.arrived;
    if (arrived > 0)
      top2_merge(top, ga.best[lane], ga.better[lane], ga.bi[lane], ga.bi2[lane], geom);
    emit = arrived == S - 1;
    if (!emit) {
      ga.best[lane] = top.best;
      ga.better[lane] = top.better;
      ga.bi[lane] = top.best_i;
      ga.bi2[lane] = top.better_i;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      ga.arrived = emit ? 0 : arrived + 1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      atomicExch(&ga.lock, 0);
    }
  }
  if (emit && active) emit_bid(c.A, c.o, j, top, c.eps);
#ifdef SN_BID_STAMPS
  STAMP(4)
  if (c.stamps && lane == 0)
    for (int i = 0; i < 16; ++i) c.stamps[i] += st[i];
#endif
}

struct AuctionArgs {
  int B, n, iters;
  float eps;
  const float *xyz1, *xyz2;
  int *assignment;
  float *dist;
  EmdWs ws;
  AuctionCtl *ctl;
  long long *stats;
  TeamGeom tg;
  int diag;  // SN_EMD_DIAG (tools/emd_ab.py): dwords[4..11] += 100 MHz ticks of team 0 / workgroup 0 per phase
             // (compact, -, bid, barrier, getmax, barrier, assign, barrier); 2: also per iteration and
             // workgroup at dwords[16 + ((it * 8 + m) * 8 + phase)].  dwords = the diag area of the workspace
             // (sn_emd_diag_offset), zeroed by the call.
  long long *dwords;
};

__global__ __launch_bounds__(kBidThreads, SN_EMD_WAVES) void emd_auction_kernel(AuctionArgs a) {
  __shared__ WaveTab tabs[kBidWaves];
  __shared__ GroupAcc gacc[kBidWaves];
  __shared__ int wsum[kBidWaves];
  __shared__ int s_flag, s_ticket, s_stray, s_range[3], s_bins[kRankBins];
  const int tid = threadIdx.x;
  if (tid < kBidWaves) {
    gacc[tid].lock = 0;
    gacc[tid].arrived = 0;
  }
  if (tid < kRankBins) s_bins[tid] = 0;
  const int G = a.tg.G;
  if (tid == 0) {
    int t = -1, stray = 0;
    if (a.tg.xcd) {
      // A team = G consecutive tickets of ONE XCD's counter: its workgroups share that XCD's L2, where the
      // cloud's streams then stay from iteration to iteration.  The XCD is read from the hardware register
      // (the ticket ORDER says nothing about placement).  A counter only hands out its share of slots; a
      // workgroup on an over-subscribed XCD walks on to the next counter -- every slot is taken whatever
      // the placement, which only ever costs speed.
      const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);  // HW_REG_XCC_ID[3:0]
      const int cap = (a.tg.teams / 8) * G;
      // SN_EMD_DIAG bit 2 (tests): ask the NEIGHBOUR XCD's counter first, so that every team is a mixed one
      for (int i = (a.diag & 4) ? 1 : 0; i < 9 && t < 0; ++i) {
        const int x = (xcc + i) & 7;
        const int k = (int)__hip_atomic_fetch_add(&a.ctl->xticket[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k < cap) {
          t = ((k / G) * 8 + x) * G + k % G;  // team * G + member
          stray = (i & 7) != 0;                // a slot of another XCD's team
        }
      }
    } else {
      t = (int)__hip_atomic_fetch_add(&a.ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_ticket = t;
    s_stray = stray;
  }
  __syncthreads();
  const int ticket = s_ticket;
  if (ticket < 0) return;  // no slot left: surplus workgroup of a grid larger than teams x G
  const int team = ticket / G, m = ticket % G;
  if (team >= a.tg.teams) return;
  TeamSync ts = {a.ctl->bar + (size_t)team * 32, &a.ctl->abort, 0u, G};
  // Is the whole team on ONE XCD?  Then its stores may stay in that XCD's L2 (see stc).  Every member that took
  // a slot of another XCD's team says so in the team's second control word; one formation barrier later every
  // member reads the same answer.  Teams that span XCDs by design (fewer than 32 clouds) never qualify.
  bool loc = false;
  if (a.tg.xcd) {
    unsigned *mixed = a.ctl->bar + (size_t)team * 32 + 1;
    if (G > 1) {
      if (tid == 0 && s_stray) __hip_atomic_fetch_or(mixed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!team_barrier(ts, &s_flag)) return;
      loc = __hip_atomic_load(mixed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    } else {
      loc = true;  // a team of one workgroup
    }
  }

  const int n = a.n, nsb = n >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int Rs = n / G, rs0 = m * Rs;  // static slice (final distances); n % 1024 == 0, G <= 64
  const int binsize = n / kRankBins;   // ranks per counter bin (a multiple of 4: n % 1024 == 0)
  const int block_cnt = n / 1024;
  float *price = a.ws.price;
  int *flags = a.ws.flags;
  int *llist_all = a.ws.list[0];
  const BidOut bo = {a.ws.bid, a.ws.bid2, a.ws.bid_inc, a.ws.max_inc, a.ws.win, loc};
  if (a.diag && m == 0 && tid == 0 && loc) atomicAdd(reinterpret_cast<unsigned long long *>(a.dwords) + 12, 1ull);  // teams on one XCD

  for (int b = team; b < a.B; b += a.tg.teams) {
    const size_t o = (size_t)b * n;
    float tmax = 0.f;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const float lo = a.ws.bbox[b * 6 + ax], hi = a.ws.bbox[b * 6 + 3 + ax];
      tmax += __builtin_fmaxf(lo * lo, hi * hi);
    }
    tmax *= 1.0001f;
    BidCtx c;
    c.n = n;
    c.nsb = nsb;
    c.eps = a.eps;
    c.tmax = tmax;
    c.o = o;
    c.p1 = a.xyz1 + o * 3;
    c.t4 = a.ws.t4s + o;
    c.pkc = a.ws.pk + o;
    c.rk2 = a.ws.rank2 + o;
    c.ms = a.ws.mstream + (size_t)b * nsb * 64;
    c.sbb = a.ws.sbbox + (size_t)b * nsb * 32;
    c.A = bo;

    for (int it = 0; it < a.iters; ++it) {
      const int cur = it & 1;
      // The unassigned bidders per 1/64 of the rank range, counted by the previous Assign: every workgroup
      // of the team derives the same split of the ranks into G contiguous, equally loaded ranges (bins are
      // indivisible).  A static split left the team waiting ~20 us per late iteration for the workgroup
      // whose region happened to hold two groups of bidders instead of one.
      if (wave == 0) {  // lane l holds the bins 4 l .. 4 l + 3
        const int2 lo2 = ldc2(a.ws.bins[cur] + b * kRankBins + 4 * lane);
        const int2 hi2 = ldc2(a.ws.bins[cur] + b * kRankBins + 4 * lane + 2);
        const int v[4] = {lo2.x, lo2.y, hi2.x, hi2.y};
        const int lsum = (v[0] + v[1]) + (v[2] + v[3]);
        int incl = lsum;
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d);
          if (lane >= d) incl += t;
        }
        const int total = __shfl(incl, 63);
        // a bin goes to the workgroup its MIDPOINT falls to in the ideal split (boundaries land on the bin edge
        // nearest to m total / G): with 256 bins a workgroup's load is within a couple of bidders of total / G,
        // so that it needs a second group of 64 only when the ideal split does
        int excl = incl - lsum, first = 0, cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int own = total > 0 ? (int)(((long long)(2 * excl + v[q]) * G) / (2LL * total)) : 0;
          own = own < G - 1 ? own : G - 1;
          first += __popcll(__ballot(own < m));
          cnt += __popcll(__ballot(own == m));
          excl += v[q];
        }
        if (lane == 0) {
          s_range[0] = total;
          s_range[1] = first * binsize;
          s_range[2] = cnt * binsize;
        }
      }
      __syncthreads();
      const int U = s_range[0], r0 = s_range[1], R = s_range[2];
      if (U == 0) break;  // every workgroup of the team reads the same value
      int *llist = llist_all + o + r0;
      const bool last = it == a.iters - 1;
      if (m == 0 && tid == 0 && a.stats) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.stats), (unsigned long long)U * n);
        if (b == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.stats) + 1, 1ULL);
      }
      // the counters Assign fills in this iteration (read last at the top of the previous one)
      if (m == 0 && tid >= 64 && tid < 64 + kRankBins)
        stc(loc, a.ws.bins[cur ^ 1] + b * kRankBins + (tid - 64), 0);
      const bool dg = a.diag && team == 0 && tid == 0;
      long long tk = dg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
      auto tick = [&](int slot) {
        if (dg) {
          const long long now = (long long)__builtin_amdgcn_s_memrealtime();
          if (m == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.dwords) + slot, (unsigned long long)(now - tk));
          if (a.diag >= 2 && m < 8 && it < 64) a.dwords[16 + ((it * 8 + m) * 8 + (slot - 4))] = now - tk;
          tk = now;
        }
      };
      // ---- local list: the raised flags of the own rank range, in rank order
      int Um = 0;
      {
        int base = 0;
        const int vec = R >> 2;
        for (int w0 = 0; w0 < vec; w0 += kBidThreads) {
          const int w = w0 + tid;
          int4 f = make_int4(0, 0, 0, 0);
          if (w < vec) {  // coherent reads (other workgroups raised these flags), two 8-byte words per lane
            const int2 lo2 = ldc2(flags + o + r0 + 4 * w), hi2 = ldc2(flags + o + r0 + 4 * w + 2);
            f = make_int4(lo2.x, lo2.y, hi2.x, hi2.y);
            if (f.x) stc(loc, flags + o + r0 + 4 * w, 0);
            if (f.y) stc(loc, flags + o + r0 + 4 * w + 1, 0);
            if (f.z) stc(loc, flags + o + r0 + 4 * w + 2, 0);
            if (f.w) stc(loc, flags + o + r0 + 4 * w + 3, 0);
          }
          const int cnt = (f.x != 0) + (f.y != 0) + (f.z != 0) + (f.w != 0);
          int incl = cnt;
          for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
          }
          if (lane == 63) wsum[wave] = incl;
          __syncthreads();
          int pos = base + incl - cnt, total = 0;
          for (int wv = 0; wv < kBidWaves; ++wv) {
            if (wv < wave) pos += wsum[wv];
            total += wsum[wv];
          }
          if (cnt > 0) {
            const int r = r0 + 4 * w;
            if (f.x) stc(loc, &llist[pos++], a.ws.perm1[o + r]);
            if (f.y) stc(loc, &llist[pos++], a.ws.perm1[o + r + 1]);
            if (f.z) stc(loc, &llist[pos++], a.ws.perm1[o + r + 2]);
            if (f.w) stc(loc, &llist[pos++], a.ws.perm1[o + r + 3]);
          }
          base += total;
          __syncthreads();
        }
        Um = base;
      }
      tick(4);
      tick(5);
      // ---- bid
      {
        c.geom = TieGeom{n, 1024 / ((U + block_cnt - 1) / block_cnt)};
        // prices in level 1 once the bidders are sparse (measured at n = 16384: from ~900 unassigned bidders
        // per cloud on the second MFMA pays for itself through fewer hits; earlier it costs 10-20 %)
        c.prt = (a.eps >= 0.f && it > 0 && U * 16 <= n) ? reinterpret_cast<const f4 *>(a.ws.prt + o) : nullptr;
        const float price_floor = a.eps < 0.f ? a.eps * (float)it : 0.f;
        c.a_max = filter_target(price_floor) + 9.5367431640625e-07f;
        const int ngroups = (Um + 63) >> 6;
        int S = 1;
        while (S < kBidWaves && S * 2 * ngroups <= kBidWaves) S *= 2;
        const int gpb = kBidWaves / S;
        const int seg = wave & (S - 1), gslot = wave / S;
#ifdef SN_BID_STAMPS
        c.stamps = (a.diag && team == 0 && m < 3 && it >= 10) ? a.dwords + 16 + 3200 + (m * 16 + wave) * 16 : nullptr;
#endif
        for (int q0 = 0; q0 < ngroups; q0 += gpb) {
          bid_group(c, tabs[wave], gacc[gslot], llist, Um, q0 + gslot, ngroups, S, seg, lane);
          if (q0 + gpb < ngroups) __syncthreads();
        }
      }
      if (a.diag) __syncthreads();
      tick(6);
      if (!team_barrier(ts, &s_flag)) return;
      tick(7);
      // ---- GetMax (emd_cuda.cu:181-194) for the own bidders
      // a thread serves the same list slots in GetMax and in Assign: what it looked up for its FIRST slot (the
      // only one once a workgroup has <= 1024 bidders) stays in registers across the barrier -- two dependent
      // round trips less at the head of Assign
      int keep_j = -1, keep_tgt = -1;
      float keep_inc = 0.f;
      for (int u = tid; u < Um; u += kBidThreads) {
        const int j = ldc(&llist[u]);
        const int tgt = ldc(&bo.bid[o + j]);
        const float bi = tgt < 0 ? 0.f : ldc(&bo.bid_inc[o + j]);
        if (u == tid) {
          keep_j = j;
          keep_tgt = tgt;
          keep_inc = bi;
        }
        if (tgt < 0) continue;
        const float mi = ldc(&bo.max_inc[o + tgt]);
        if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6)
          __hip_atomic_fetch_max(&bo.win[o + tgt], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (a.diag) __syncthreads();
      tick(8);
      if (!team_barrier(ts, &s_flag)) return;
      tick(9);
      // ---- Assign (emd_cuda.cu:196-215) for the own bidders; raised flags are counted for the next U
      {
        unsigned *nextbins = reinterpret_cast<unsigned *>(a.ws.bins[cur ^ 1] + b * kRankBins);
        auto raise = [&](int rank) {  // counted per bin in LDS first: <= 64 device atomics per workgroup
          stc(loc, &flags[o + rank], 1);
          atomicAdd(&s_bins[rank / binsize], 1);
        };
        for (int u = tid; u < Um; u += kBidThreads) {
          const bool kept = u == tid;
          const int j = kept ? keep_j : ldc(&llist[u]);
          const int tgt = kept ? keep_tgt : ldc(&bo.bid[o + j]);
          if (tgt < 0) {  // no bid (non-finite input): stays unassigned, distance 0, zero gradient
            if (!last) raise(a.ws.rank1[o + j]);
            continue;
          }
          // GetMax only writes max_idx when some bidder's increment is within 1e-6 of max_increments, and the
          // reference never clears that tensor (emd_cuda.cu:181-194, emd_module.py:50: zeros): when nobody is
          // in the window -- max_increments still holds its initial 0 and every increment is negative, i.e.
          // eps < 0 -- Assign compares against the entry of an EARLIER iteration (initially 0).  Every bidder
          // of a target sees the same `win`, so they all take the same branch: no read races a write.
          int w = ldc(&bo.win[o + tgt]);
          if (w >= 0)
            stc(loc, &a.ws.max_idx[o + tgt], w);
          else
            w = ldc(&a.ws.max_idx[o + tgt]);
          if (last || w == j) {
            const int inv = ldc(&a.ws.assignment_inv[o + tgt]);
            if (!last && inv != -1) {
              stc(loc, &a.assignment[o + inv], -1);
              raise(a.ws.rank1[o + inv]);  // evicted: bids again
            }
            stc(loc, &a.ws.assignment_inv[o + tgt], j);
            stc(loc, &a.assignment[o + j], tgt);
            const float np = ldc(&price[o + tgt]) + (kept ? keep_inc : ldc(&bo.bid_inc[o + j]));
            stc(loc, &price[o + tgt], np);
            const int pos = a.ws.rank2[o + tgt];  // the bid phase reads the price by stream position
            stc(loc, reinterpret_cast<float *>(a.ws.pk + o + pos), np);
            stc(loc, a.ws.prt + o + ((pos >> 6) * 64 + (pos & 15) * 4 + ((pos >> 4) & 3)), np);  // [sb][c][q]
            stc(loc, &bo.max_inc[o + tgt], -1e9f);
          } else {
            raise(a.ws.rank1[o + j]);  // lost: bids again
          }
        }
        __syncthreads();
        if (tid < kRankBins) {
          const int v = s_bins[tid];
          if (v) {
            __hip_atomic_fetch_add(nextbins + tid, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_bins[tid] = 0;
          }
        }
      }
      if (a.diag) __syncthreads();
      tick(10);
      if (!team_barrier(ts, &s_flag)) return;
      tick(11);
    }
    // ---- distances of the final assignment (emd_cuda.cu:218-226), own slice of the bidder indices
    {
#pragma clang fp contract(off)
      for (int e = rs0 + tid; e < rs0 + Rs; e += kBidThreads) {
        const int k = ldc(&a.assignment[o + e]);
        float d = 0.f;
        if (k >= 0) {
          const float *p = a.xyz1 + (o + e) * 3, *q = a.xyz2 + (o + k) * 3;
          const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
          d = (dx * dx + dy * dy) + dz * dz;
        }
        a.dist[o + e] = d;
      }
    }
    // a team that serves several clouds: the flags / lists of the next cloud are its own, nothing to wait for
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
    const int k = assignment[e];
    if (k < 0) {  // unassigned (iters == 0 or non-finite input)
      grad[e * 3 + 0] = grad[e * 3 + 1] = grad[e * 3 + 2] = 0.f;
      continue;
    }
    const float *a = xyz1 + e * 3, *o = xyz2 + (bb * n + k) * 3;
    const float g = graddist[e] * 2;
    grad[e * 3 + 0] = g * (a[0] - o[0]);
    grad[e * 3 + 1] = g * (a[1] - o[1]);
    grad[e * 3 + 2] = g * (a[2] - o[2]);
  }
}

constexpr size_t kDiagWords = 16 + 64 * 64;                        // int64 phase timers (SN_EMD_DIAG)
constexpr size_t kCtlWords = 32 + 32 * 1024;                        // ticket, abort, up to 1024 team counters
constexpr size_t kCtlBytes = 4 * kCtlWords + 8 * kDiagWords;

EmdWs carve(void *workspace, int b, int n) {
  char *p = static_cast<char *>(workspace);
  const size_t arr = sn::align_up((size_t)b * n * 4, 256);
  EmdWs ws;
  ws.assignment_inv = reinterpret_cast<int *>(p); p += arr;
  ws.price = reinterpret_cast<float *>(p); p += arr;
  ws.bid = reinterpret_cast<int *>(p); p += arr;
  ws.bid2 = reinterpret_cast<int *>(p); p += arr;
  ws.bid_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_idx = reinterpret_cast<int *>(p); p += arr;
  ws.win = reinterpret_cast<int *>(p); p += arr;
  ws.list[0] = reinterpret_cast<int *>(p); p += arr;
  ws.prt = reinterpret_cast<float *>(p); p += arr;
  ws.bins[0] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * kRankBins * 4, 256);
  ws.bins[1] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * kRankBins * 4, 256);
  ws.t4s = reinterpret_cast<f4 *>(p); p += sn::align_up((size_t)b * n * 16, 256);
  ws.pk = reinterpret_cast<float2 *>(p); p += sn::align_up((size_t)b * n * 8, 256);
  ws.rank2 = reinterpret_cast<int *>(p); p += arr;
  ws.mstream = reinterpret_cast<f4 *>(p); p += sn::align_up((size_t)b * n * 16, 256);
  ws.tperm = reinterpret_cast<int *>(p); p += arr;
  ws.cell_of = reinterpret_cast<int *>(p); p += arr;
  ws.hist = reinterpret_cast<int *>(p); p += (size_t)b * kSortCells * 4;
  ws.bbox = reinterpret_cast<float *>(p); p += sn::align_up((size_t)b * 24, 256);
  ws.sbbox = reinterpret_cast<float *>(p); p += sn::align_up((size_t)b * (n / 16) * 32, 256);
  ws.perm1 = reinterpret_cast<int *>(p); p += arr;
  ws.rank1 = reinterpret_cast<int *>(p); p += arr;
  ws.flags = reinterpret_cast<int *>(p); p += arr;
  ws.hist1 = reinterpret_cast<int *>(p); p += (size_t)b * kSortCells * 4;
  ws.bbox1 = reinterpret_cast<float *>(p); p += sn::align_up((size_t)b * 24, 256);
  ws.ctl = p; p += kCtlBytes;
  return ws;
}

}  // namespace

extern "C" size_t sn_emd_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  return 16 * sn::align_up((size_t)b * n * 4, 256) + 2 * sn::align_up((size_t)b * kRankBins * 4, 256) +
         2 * sn::align_up((size_t)b * n * 16, 256) + sn::align_up((size_t)b * n * 8, 256) + 2 * (size_t)b * kSortCells * 4 +
         2 * sn::align_up((size_t)b * 24, 256) + sn::align_up((size_t)b * (n / 16) * 32, 256) + kCtlBytes;
}

// byte offset of the diagnostic words (SN_EMD_DIAG) inside the workspace
extern "C" size_t sn_emd_diag_offset(int b, int n) {
  return sn_emd_workspace_bytes(b, n) - 8 * kDiagWords;
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
  SN_REQUIRE(cloud_sort(b, n, xyz2, ws.bbox, ws.hist, ws.cell_of, ws.tperm, s) == 0,
             "sn_emd_forward: cannot size the sort kernel's LDS");
  SN_REQUIRE(cloud_sort(b, n, xyz1, ws.bbox1, ws.hist1, ws.cell_of, ws.perm1, s) == 0,
             "sn_emd_forward: cannot size the sort kernel's LDS");
  static const bool check = [] { const char *e = getenv("SN_EMD_CHECK"); return e && e[0] == '1'; }();
  emd_init_kernel<<<eblocks, kThreads, 0, s>>>(b, n, xyz2, assignment, ws);
  emd_sbbox_kernel<<<(int)(((long)b * (n / 64) + 3) / 4), 256, 0, s>>>(b, n, xyz2, ws);
  emd_seed_kernel<<<eblocks, kThreads, 0, s>>>(b, n, xyz1, xyz2, ws);
  {
    int dev = 0, cus = 0;
    SN_HIP(hipGetDevice(&dev));
    SN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    SN_REQUIRE(cus >= 1 && cus <= 1024, "sn_emd_forward: unexpected compute-unit count %d", cus);
    // one workgroup of 16 waves per CU (the register budget admits exactly one): the whole grid is resident
    // on an idle device, and the ticket order keeps it live next to other launches (see the kernel's header)
    AuctionArgs args;
    args.B = b;
    args.n = n;
    args.iters = iters;
    args.eps = eps;
    args.xyz1 = xyz1;
    args.xyz2 = xyz2;
    args.assignment = assignment;
    args.dist = dist;
    args.ws = ws;
    args.ctl = static_cast<AuctionCtl *>(ws.ctl);
    args.stats = stats;
    args.tg = team_geometry(b, cus);
    static const int diag = [] { const char *e = getenv("SN_EMD_DIAG"); return e ? atoi(e) : 0; }();
    args.diag = diag;
    args.dwords = reinterpret_cast<long long *>(static_cast<char *>(ws.ctl) + 4 * kCtlWords);
    if (diag) SN_HIP(hipMemsetAsync(args.dwords, 0, 8 * kDiagWords, s));
    SN_HIP(hipMemsetAsync(ws.ctl, 0, 4 * (32 + 32 * (size_t)args.tg.teams), s));
    SN_TIMED("emd_auction", s, (emd_auction_kernel<<<cus, kBidThreads, 0, s>>>(args)));
    if (check) {  // debugging aid: a barrier that timed out leaves garbage in dist / assignment
      unsigned abort_word = 0;
      SN_HIP(hipStreamSynchronize(s));
      SN_HIP(hipMemcpy(&abort_word, &args.ctl->abort, 4, hipMemcpyDeviceToHost));
      SN_REQUIRE(abort_word == 0, "sn_emd_forward: a team barrier of the persistent auction timed out");
    }
    return sn::launch_status("sn_emd_forward");
  }
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
