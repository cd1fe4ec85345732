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
//   * GetMax + Assign are ONE target-centric phase: the bid epilogue links the bidders of a target into a list
//     (atomic exchange on the target's head word); after ONE team barrier the list's head walks it, applies the
//     reference's +-1e-6 window and its highest-bidder-index rule, awards the target and re-flags the losers.
//     Two team barriers per iteration instead of three, no separate GetMax pass.
//   * What the auction costs depends on the DATA (round 5; tools/emd_regimes.py, bench.py `emd_regimes`): uniform cubes
//     are its easiest input.  A prediction that lies OFF the targets' surface (counted by the seed kernel) bids with
//     the per-bidder scan from the first iteration on, its workgroups own a transposed comb of rank bins instead of a
//     contiguous range (the cost per bidder is what varies there) and the scan re-tightens its reach between lists;
//     a CONTESTED auction (hundreds of bidders per target: the refine stages of an untrained generator) is noticed by
//     the award phase's walkers, and from the next iteration on a bidder that finds its target's running maximum
//     already above its own increment does not enter the target's list at all.  All of it exact: which workgroup
//     serves a bidder, in which order, and whether a certain loser is linked never enter a result.
//   * A raised flag IS the bidder's index (+ 1) and the index travels in the bid record: neither the compaction nor
//     the award phase looks anything up in the permutation.
//   * Round 6: in the per-bidder scan a box of 16 (or 256) targets is tested against a reach computed from the box's OWN
//     price bound (box_within_priced); on contested clouds that bound is the smallest price the box holds, refreshed
//     in LDS every fourth contested iteration -- a far bidder no longer lists the expensive near-side blocks.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>

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
// (Round 6 measured WORKGROUP-scope atomics for teams that sit on one XCD -- the running maxima, the list heads, the bin
// counters and the barrier arrivals, on the argument that the team meets in that XCD's L2 exactly as its plain stores do:
// bit-exact, verified by a third litmus part, and 2-3 % SLOWER on every kind of data (uniform 2.106 against 2.037 ms per
// call at 32 clouds, untrained 31.6 against 30.8; profiles/r06_e_emd_atomics_scope_not_kept.txt).  Agent scope stays.)
__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
  if (v >= 0.f)
    __hip_atomic_fetch_max(reinterpret_cast<int *>(addr), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_fetch_min(reinterpret_cast<unsigned *>(addr), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same, returning the value the word held before (NaN-free inputs; -1e9 and 0 are the only other values stored)
__device__ __forceinline__ float atomic_max_float_old(float *addr, float v) {
  if (v >= 0.f)
    return __int_as_float(__hip_atomic_fetch_max(reinterpret_cast<int *>(addr), __float_as_int(v), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT));
  return __uint_as_float(__hip_atomic_fetch_min(reinterpret_cast<unsigned *>(addr), __float_as_uint(v), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT));
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
__device__ __forceinline__ void stc2(bool loc, int *p, int x, int y) {  // 8-byte aligned pair, one store
  const unsigned long long v = (unsigned long long)(unsigned)x | ((unsigned long long)(unsigned)y << 32);
  if (loc)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc64(bool loc, unsigned long long *p, unsigned long long v) {
  if (loc)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ldc64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  int *rec;      // [B, n, 4] by bidder RANK: {bid increment bits, rank of the next bidder of the same target or -1,
                 //  bidder index, -}: one 16-byte line segment per step of the award phase's walk
  float *max_inc;
  int *max_idx;  // GetMax's winner per target (as a bidder RANK); PERSISTS across iterations like the reference's tensor
  int *head;     // [B, n] per target: rank of the bidder that pushed last in this iteration's bid phase, -1: no bid
  int *list[1];  // [B, n, 2] the unassigned bidders of the iteration as {index, rank}, per workgroup in its own rank range
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
  int *far;      // [B] bidders that lie far from their nearest seed (emd_seed_kernel; nullptr: not counted)
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
    ws.head[e] = -1;
    ws.bid[e] = -1;   // no previous favourites yet (filter seeding)
    ws.bid2[e] = -1;
    ws.rank1[e - e % n + ws.perm1[e]] = (int)(e % n);  // max_idx: see emd_seed_kernel (needs rank1 complete)
    // every bidder starts flagged (= unassigned).  A raised flag IS the bidder's index + 1 (by rank): the compaction
    // builds its {index, rank} list from the flag words alone, without a dependent look-up in perm1
    ws.flags[e] = ws.perm1[e] + 1;
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
    if (e < B) ws.far[e] = 0;
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
// tighten.  ANY two distinct targets are valid seeds; better ones only make the filter reject more.
// Targets are already in Morton order: a window of 16 sorted positions around the
// bidder's own cell supplies candidates, the two nearest become bid / bid2.
// The kernel also COUNTS (on a sample of the waves) the bidders that lie far from the targets: a prediction off
// the targets' surface is data on which the auction bids with the scan from the first iteration on (see the
// kernel's bid phase).  (Round 5 also built seeds from the box hierarchy -- the two nearest groups of 256 targets,
// their nearest blocks, the two nearest targets of those: better seeds for far bidders, 13 us less in the first
// iteration of the scattered probe; once such clouds bid with the scan it made no difference any more -- 3.98 vs
// 3.99 ms per call -- and the kernel cost 40 us of LDS fills on every cloud: removed.)
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
    // the reference's max_idx tensor starts as zeros = "bidder 0" (emd_module.py:50); kept here as a RANK
    ws.max_idx[e] = ws.rank1[bb * n];
    // Is this bidder FAR from the targets -- farther from its nearest seed than a quarter of the width of the block of
    // 16 stream neighbours it looked at?  (uniform cubes: 8 % of the bidders, prediction = ground truth + 1 % noise:
    // none, scattered +-0.3 around the surface: 58 %, the refine stages of an untrained generator: 85 %.)  Counted on
    // every eighth wave, scaled by 8: one atomic per wave on the cloud's counter took 58 us at 32 clouds (256
    // serialised atomics per word), an eighth of them is an estimate that is good enough for a 25 % threshold.
    // (which eighth: ONE wave of every eight consecutive ones, at an offset that rotates from group to group --
    // exactly n / 512 samples per cloud, spread over the Hilbert curve instead of the first wave of every group)
    const unsigned wv = (unsigned)((e - bb * n) >> 6);
    if (((wv * 5u) & 7u) == ((wv >> 3) & 7u)) {  // wave-uniform (n % 1024 == 0: a wave never straddles two clouds)
      const float *bx = ws.sbbox + (bb * (n >> 4) + ((lo + 8) >> 4)) * 8;
      const float ex = bx[3] - bx[0], ey = bx[4] - bx[1], ez = bx[5] - bx[2];
      const unsigned long long farm = __ballot(16.f * s1 > (ex * ex + ey * ey) + ez * ez);
      if ((threadIdx.x & 63) == 0 && farm != 0ull) atomicAdd(&ws.far[bb], 8 * __popcll(farm));
    }
  }
}

struct BidOut {
  int *bid, *bid2;
  int *rec;
  float *max_inc;
  int *head;
  bool loc;  // the team sits on one XCD: plain stores (see stc)
  int skip;  // outbid-skip (see emit_bid): 0 = off
};

#ifndef SN_EMD_BIDWAVES
#define SN_EMD_BIDWAVES 16   // waves per workgroup; 16 / SN_EMD_BIDWAVES workgroups share a CU
#endif
constexpr int kBidWaves = SN_EMD_BIDWAVES;
constexpr int kBidThreads = kBidWaves * 64;
#ifdef SN_EMD_WG_PER_CU   // experiment builds: e.g. ONE 8-wave workgroup per CU, half of every SIMD's registers left free
constexpr int kWgPerCu = SN_EMD_WG_PER_CU;
#else
constexpr int kWgPerCu = 16 / kBidWaves;
#endif
constexpr int kStash = kBidThreads;  // list slots whose bid is handed to the award phase through LDS

// What the award phase needs to know about the bid of list slot u (written by the wave that emits the bid, read
// after the team barrier by thread u): saves the llist -> bid -> rec chain of dependent coherent loads.
// next == kOutbid (also in rec[rank].next): the bidder found itself outbid on arrival and did not link itself
// (emit_bid): the award phase only re-flags it.
struct BidStash {
  int tgt, rank, inc_bits, next, j;
};
constexpr int kOutbid = -3;

// The bid of bidder j (Morton rank `rank`, list slot u).  Besides the favourites (next iteration's filter seeds)
// and the running maximum of the target's increments (emd_cuda.cu:175-177), the bidder LINKS itself into the
// list of its target: head[target] <- rank, rec[rank] = {increment, previous head}.  After the team barrier the
// bidder that finds itself at the head walks the list (award phase of the kernel).
//
// Outbid on arrival (A.skip).  GetMax's winner lies inside the window |inc - mi| <= 1e-6 around the FINAL maximum mi
// of the target's increments (emd_cuda.cu:188), and the running maximum only grows inside an iteration: a bidder
// that finds the word already above inc + 1e-6 (the reference's double compare) can never be inside the window,
// whatever arrives later.  For eps >= 0 somebody always IS inside it (the bidder that sets the maximum; the word
// starts at 0 <= every increment), so the persistent max_idx entry never decides and such a bidder simply stays
// unassigned (emd_cuda.cu:200: neither forced nor the winner): it does not enter the list; its record says so and
// the award phase re-flags it (there, not here: the flag array, the bin counters and the bin size would otherwise
// be live through the whole bid phase of a kernel that sits at its register limit -- 13 more spilled VGPRs).  With L bidders on one target arriving in random order ~ln L of them link themselves: on the refine stages
// of an untrained generator (hundreds of far bidders on every near target: lists of 200-480, walked by ONE thread
// at a dependent load per entry) 7000 bidders leave 1400 list entries, the longest list 10
// (tools/sim/auction_regime_stats.c).  Costs one more dependent round trip per bid (the maximum must RETURN the old
// value before the exchange), so it is switched on per iteration by the lists the award phase walked in the
// previous one (the team's `cont` word).  Never in the last iteration (every bidder takes its target there) and
// never for eps < 0.
__device__ __forceinline__ void emit_bid(const BidOut &A, size_t o, int j, int rank, int u, BidStash *stash,
                                         const Top2 &top, float eps) {
  const bool loc = A.loc;
  const bool st = u < kStash;
  if (top.best_i < 0) {  // only with non-finite coordinates: no comparison succeeded
    stc(loc, &A.bid[o + j], -1);
    stc(loc, &A.bid2[o + j], -1);
    stc2(loc, &A.rec[4 * (o + rank)], 0, -1);
    stc(loc, &A.rec[4 * (o + rank) + 2], j);
    if (st) stash[u] = BidStash{-1, rank, 0, -1, j};
    return;
  }
  const float inc = (top.best - top.better) + eps;
  stc(loc, &A.bid[o + j], top.best_i);
  stc(loc, &A.bid2[o + j], top.better_i == top.best_i ? -1 : top.better_i);
#ifdef SN_EMD_NOSKIP
  if (false) {
#else
  if (A.skip) {
#endif
    const float before = atomic_max_float_old(&A.max_inc[o + top.best_i], inc);
    if ((double)before > (double)inc + 1e-6) {  // outbid already: stays unassigned, bids again
      stc2(loc, &A.rec[4 * (o + rank)], __float_as_int(inc), kOutbid);
      stc(loc, &A.rec[4 * (o + rank) + 2], j);
      if (st) stash[u] = BidStash{top.best_i, rank, __float_as_int(inc), kOutbid, j};
      return;
    }
  } else {
    atomic_max_float(&A.max_inc[o + top.best_i], inc);
  }
  const int prev = (int)__hip_atomic_exchange(reinterpret_cast<unsigned *>(&A.head[o + top.best_i]), (unsigned)rank,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  stc2(loc, &A.rec[4 * (o + rank)], __float_as_int(inc), prev);
  stc(loc, &A.rec[4 * (o + rank) + 2], j);  // the bidder's index travels with its rank: no perm1 look-up in the award phase
  if (st) stash[u] = BidStash{top.best_i, rank, __float_as_int(inc), prev, j};
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
  int sj[64], sr[64];
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
// the whole call and walks  compact -> bid -> [barrier] -> award (GetMax + Assign) -> [barrier]  with a team barrier
// (monotonic counter; the exchanged words are read and written with coherent accesses: placement
// independent) between the phases: TWO barriers per iteration.
//   * The Morton RANKS of the bidders are split among the team's workgroups at the top of every iteration
//     (contiguous ranges of equal load, from the per-bin counters the previous award phase left): a workgroup
//     compacts the raised flags of its range into a local list (no global scan), bids for those bidders and
//     walks the target lists whose head it emitted.  Its bidders stay spatial neighbours, and the targets near
//     them stay in its L1 / the team's L2 from one iteration to the next.  The order in which bidders are served
//     never enters a result (top-2 values, tie keys, atomicMax winners), so this is bit-identical.
//   * S = 2^k <= 16 waves share one local group of 64 bidders exactly as above (dense iterations); sparse
//     iterations serve every bidder with a quarter wave (bid_scan).
//   * Teams are formed from a TICKET taken at start, not from blockIdx: a workgroup only ever waits for
//     workgroups that have started, and at most one team per ticket counter is incomplete at any time, so ordinary
//     kernels on other streams only delay a launch (they finish and free their CUs).  Two TEAM-WAITING launches of
//     this process never overlap: sn::PersistentLaunch chains them through an event (common.hpp) -- between them
//     they could otherwise hold every CU of an XCD with members of incomplete teams.
//     A team is G consecutive tickets of ONE XCD's counter for every batch size (the XCD read from the hardware
//     register: its L2 then keeps the cloud's streams and the team's stores; speed only); devices whose
//     workgroup count is not a multiple of 64 get contiguous teams from one global counter.
//   * Every spin is bounded; a time-out raises ctl.abort and the device's sticky word, every workgroup of the
//     launch leaves, the unfinished clouds get dist = NaN / assignment = -1, and the NEXT sn_emd_* / sn_mds call
//     on the device -- or sn_device_status() at the caller's own sync point -- returns SN_ETIMEDOUT
//     (SN_EMD_CHECK=1: the failing call itself synchronises and reports).
// =======================================================================================
struct AuctionCtl {  // zeroed by a memset node before every launch
  unsigned ticket;
  unsigned abort;
  unsigned xticket[8];  // one ticket counter per XCD (teams of the XCD-local geometry)
  // measurement aid (sn_prof_enable): the launch's own execution window in 100 MHz ticks -- ~(earliest start of a
  // working workgroup) and the latest end, both as running maxima so that the zeroed block is their neutral element
  unsigned long long t_first_inv, t_last;
  unsigned pad[18];
  unsigned bar[1];  // [teams * 32]: one counter per team, 128 bytes apart
};

// Two words per team (zeroed by the memset node before every launch): what one iteration tells the next.
//   cont[parity] = stamp of the iteration that should treat the auction as CONTESTED (hundreds of bidders per
//   target): its bidders check the target's running maximum before they link themselves (emit_bid's outbid-skip) and
//   every workgroup bids with the scan.  Stamps grow from cloud to cloud and from iteration to iteration, so a word
//   left by an earlier cloud or iteration never looks like this iteration's.
// (Round 5 also built a work HAND-OFF on top of such a block -- a workgroup published its list, waves of workgroups
// that had finished took chunks of it through a per-workgroup cursor: bit-exact, and slower on every kind of data:
// hundreds of idle waves converge on the cursor of the one slow workgroup, whose own waves then queue behind them for
// every chunk; the iteration of the scattered probe got 70 % longer while the thieves were busy
// (profiles/r05_a_emd_handoff_not_kept.txt).  What balances the team instead is WHICH ranks a workgroup owns: see
// the split at the top of an iteration.)
// The words live in a 128-byte block of their own per team, BEHIND the teams' barrier blocks (AuctionCtl::bar + 32
// (teams + team): words 0-1 = cont[2]) and are zeroed by the same memset node: the barrier block's line is the one
// thread 0 of every workgroup of the team polls, a store to it bounces that line between the pollers.  Stamps are
// full 32-bit words (cloud_seq (iters + 1) + it + 1 cannot wrap: b <= 512 clouds, the stamp of an ALIASED word would
// only force the contested mode, which is exact too).
constexpr int kNoteWord = 0;

struct TeamGeom {
  int G;       // workgroups per team (power of two)
  int teams;   // teams in the launch
  int xcd;     // 1: a team = G consecutive tickets of one XCD's counter (XCD-local teams)
};

// W workgroups (one per CU) are split into teams of G; team t serves the clouds t, t + teams, ...
//   * XCD-local geometry (W a multiple of 64: 8 XCDs x W/8 CUs): a team lives on ONE XCD -- its barrier counter,
//     flags, bids and prices stay in that XCD's L2 (plain stores, see stc) and a barrier costs ~1.1 us.  Each XCD
//     hosts ceil(B / 8) teams of G = 2^k <= (W/8) / ceil(B/8) workgroups.  With fewer than 8 clouds some XCDs
//     idle: a late iteration is bound by the per-workgroup latency chain, not by the number of workgroups, and
//     the cross-XCD team of 64 it replaces paid 2.7 us per barrier and coherent (fabric) stores.
//   * otherwise (other partition modes, small devices): contiguous teams from one global counter.
// gmax: upper bound on G (SN_EMD_G, experiments); legacy != 0: round 2's geometry (SN_EMD_GEOM=1).
__host__ __device__ inline TeamGeom team_geometry(int B, int W, int gmax = 64, int legacy = 0) {
  TeamGeom t;
  const bool xcd_ok = W >= 64 && W % 64 == 0;
  if (xcd_ok && (B >= 32 || !legacy)) {
    const int per = W / 8;                 // workgroups of one XCD
    const int tpx = (B + 7) / 8;           // teams an XCD must host so that every cloud has its own
    int g = 1;
    while (g * 2 * tpx <= per && g * 2 <= gmax) g *= 2;
    if (W < 128) g = 1;  // see below: concurrent launches on a small device
    t.G = g;
    t.teams = (per / g) * 8;
    t.xcd = 1;
  } else {
    int g = 1;
    while (g < 64 && g * 2 * B <= W && g * 2 <= gmax) g *= 2;
    // Two launches on two streams can each hold the members of one incomplete team per ticket counter waiting for
    // the rest (up to G - 1 compute units each): with fewer than 2 x 63 compute units that could be all of them, so
    // small devices / partitions get teams of one workgroup, which wait for nobody.
    if (W < 128) g = 1;
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
  unsigned *sticky;  // the device's sticky error word (host-visible; sn_emd_* report it at their next call)
  unsigned target;   // arrivals expected at the next barrier
  unsigned limit;    // spins before a barrier gives up
  int G;
  int fenced;        // 1: agent-scope release / acquire around the barrier (SN_EMD_SAFE or a failed self-test)
};

// All waves of all G workgroups arrive.  No fence by default: every word a phase hands to the next one is written
// and read through the coherent accessors above (ldc / stc / atomics); each wave drains its stores
// (s_waitcnt vmcnt(0)) before its workgroup arrives, the arrival counter is a device-scope atomic.
// Measured: 1.1 us per barrier instead of 3.7 us with an agent-scope release + acquire pair, and the phases
// after it no longer start with an invalidated L1 / L2.  This relies on gfx950's cache behaviour (write-through
// L1, sc1 accesses meeting in the L2 / at the fabric), which the HIP memory model does not promise: the library
// verifies it once per device with a litmus kernel (emd_litmus_kernel) and falls back to `fenced` barriers +
// agent-scope stores when the check fails or SN_EMD_SAFE=1 is set.
__device__ __forceinline__ bool team_barrier(TeamSync &ts, int *s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ts.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
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
        if (spins > ts.limit) {
          __hip_atomic_store(ts.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ts.sticky) __hip_atomic_store(ts.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          ok = 0;
          break;
        }
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  if (ts.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return *s_flag != 0;
}

struct BidCtx {
  int n, nsb;
  float eps, a_max, tmax;
  TieGeom geom;
  size_t o;
  const float *p1;   // bidders of this cloud
  const float *p2;   // targets of this cloud by INDEX (the caller's array)
  const float *price;  // prices by target index (coherent reads)
  const f4 *t4;      // by stream position; prices inside change between the phases (no __restrict__)
  const float2 *pkc;
  const int *rk2;
  const f4 *ms;      // MFMA operand stream of this cloud
  const f4 *prt;     // prices of this cloud, transposed (plain loads on purpose: see below); nullptr: not used
  const float *sbb;  // block boxes of this cloud
  BidOut A;
  BidStash *stash;   // LDS, kStash entries
  bool offsurf;      // far or contested data: bid_scan tightens its reach between lists
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
    const int2 jr = ldc2(&lst[2 * (active ? u : grp * 64)]);  // {bidder index, Morton rank}
    const int jj = jr.x;
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
    ga.sr[lane] = jr.y;
  }
  // every wave of the workgroup calls bid_group the same number of times.  (With S == 1 the hand-over is private to the
  // wave and the barrier is not needed for correctness; without it the waves drift apart and the call is 1.3 % slower.)
  __syncthreads();
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
  if (emit && active) emit_bid(c.A, c.o, j, ga.sr[lane], u, c.stash, top, c.eps);
#ifdef SN_BID_STAMPS
  STAMP(4)
  if (c.stamps && lane == 0)
    for (int i = 0; i < 16; ++i) c.stamps[i] += st[i];
#endif
}

// ---------------------------------------------------------------------------------------
// Bid phase of a SPARSE iteration: one quarter wave (16 lanes) per bidder.
//
// Once a workgroup has at most SN_EMD_SCAN_MAX unassigned bidders they are far apart in their rank range: a
// group of 16 of them spans a third of the cloud, the union of their reaches (what bid_group's waves visit with the
// matrix cores) is several times what any ONE of them can reach, and an iteration is a chain of dependent steps
// (box tests -> ~9 visits -> hit queue -> merge of 16 segments) whose length, not whose work, sets the time.
// Here every bidder is served on its own:
//  * set-up, two round trips: the list entry waits in LDS (left in the slot's stash by the compaction), the two
//    previous favourites' coordinates and prices are read BY INDEX (the caller's array, the award phase's price
//    array); their values give bid_group's bound `cm` and the reach r2 (the same formulas);
//  * the reach against the boxes of the groups of 16 blocks (256 targets), then of the blocks of 16 targets of
//    the groups within reach -- all in LDS (ScanLds: the cloud's sbbox rows, copied once per cloud), four box
//    tests per lane and one per group within reach; the blocks within reach go to a list in LDS, the blocks of
//    the two previous favourites first;
//  * the listed blocks' targets, four blocks per round, lane c = target c of each: coordinates + price of the
//    next round are in flight while a round goes through the precise filter; what passes (~7 targets per bidder)
//    takes the reference's arithmetic at once; after the FIRST round (which holds the favourites' blocks) the
//    lanes share the second largest value they have seen, a bound that is final in most cases;
//  * the 16 lanes' top-2's meet through four DPP steps.
// With few bidders left 2 or 4 quarter waves share a bidder (its blocks dealt out in turn), so that a workgroup
// with 16 bidders -- the 4-clouds-per-GPU share of an 8-GPU job -- still uses all of its lanes.
// No matrix cores, no queue, no LDS election, no merge across waves, no __syncthreads.
// The same three facts make it exact: a target outside the reach cannot pass the precise filter, a target that
// fails the precise filter has a value below a proven lower bound of the bidder's final `better` (a stale or
// smaller bound only lets more through), and top2_push / top2_merge give the full scan's values and canonical
// index in any order.
// ---------------------------------------------------------------------------------------
#ifndef SN_EMD_SCAN_MAX
#define SN_EMD_SCAN_MAX 384  // bidders per workgroup up to which an iteration takes bid_scan (SN_EMD_SCAN overrides;
                             // 128 / 256 / 384 / 512 / 1024: 2.24 / 2.10 / 2.07 / 2.07 / 2.08 ms per call at 32 clouds, r04)
#endif
constexpr int kScanContested = 4096;  // bidders per workgroup up to which a CONTESTED iteration takes bid_scan (see the kernel)
constexpr int kScanBlk = 1024;  // blocks of 16 targets whose boxes fit in the LDS copy (n <= 16384)
constexpr int kScanList = 64;   // blocks within reach a quarter wave lists before it evaluates them

struct ScanLds {
  f4 hb[kScanBlk / 16][2];  // boxes of the groups of 16 blocks (256 targets), laid out like a row of sbbox
  f4 blk[kScanBlk][2];      // the cloud's sbbox rows
  unsigned short list[kBidWaves][4][kScanList];
};

__device__ __forceinline__ bool box_within(const f4 A, const f4 B, float x, float y, float z, float r2) {
  const float gx = __builtin_fmaxf(__builtin_fmaxf(A.x - x, x - A.w), 0.f);
  const float gy = __builtin_fmaxf(__builtin_fmaxf(A.y - y, y - B.x), 0.f);
  const float gz = __builtin_fmaxf(__builtin_fmaxf(A.z - z, z - B.y), 0.f);
  return ((gx * gx + gy * gy) + gz * gz) * 0.9999f <= r2;
}
// The same with the box's OWN price bound (round 6).  B.z = A'max of the box: an upper bound of filter_target(price) over
// its targets -- from the price floor (the same for every box) or, on contested / off-surface clouds, from the smallest
// price the box held when the iteration began (auction_body refreshes the LDS copy; prices do not move during a bid
// phase).  The reach is bid_group's coarse_threshold with that bound: cthr = filter_thr(bound of the bidder's final
// `better`), base = the slack term.  A bidder far from every cheap target no longer lists the expensive near-side
// blocks it cannot want: 20-35 % fewer blocks late in a contested call (tools/sim/auction_regime_stats.c).
__device__ __forceinline__ bool box_within_priced(const f4 A, const f4 B, float x, float y, float z, float cthr, float base) {
  const float r = B.z - cthr;
  const float v = __builtin_fmaf(r * __builtin_fabsf(r), 1.00000095367431640625f, base);
  return box_within(A, B, x, y, z, v > 0.f ? v * 1.0001f : v);
}

#define SN_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xf, 0xf, true))
#define SN_DPP_I(v, ctrl) __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true)

#ifndef SN_EMD_BMIN_EVERY
#define SN_EMD_BMIN_EVERY 4   // contested iterations between two refreshes of the boxes' price bounds
#endif
#ifndef SN_SCAN_RESHARE
#define SN_SCAN_RESHARE 0  // > 0: the lanes share their bound again every that many rounds (see bid_scan)
#endif
constexpr int kRoundC = 4;  // blocks per round (two rounds are in flight: the register budget decides)
struct ScanCand {  // one round of a quarter wave: lane c holds target c of each of the round's blocks
  f4 t[kRoundC];     // {x, y, z, index bits}
  float p[kRoundC];  // today's price
};


__device__ __forceinline__ void bid_scan(const BidCtx &c, ScanLds &SL, const int *lst, int count, int wave,
                                         int lane) {
  const int row = lane >> 4, col = lane & 15;
  const int nh = c.nsb >> 2;  // groups of 16 blocks
  const int nblk = c.nsb << 2;
  const TieGeom geom = c.geom;
  const f4 *t4 = c.t4;
  const float2 *pkc = c.pkc;
  unsigned short *lq = SL.list[wave][row];
#undef STAMP
#undef COUNT
#ifdef SN_BID_STAMPS
  long long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tk = (long long)__builtin_amdgcn_s_memrealtime();
#define STAMP(i) { const long long now_ = (long long)__builtin_amdgcn_s_memrealtime(); st[i] += now_ - tk; tk = now_; }
#define COUNT(i, v) st[i] += (v);
#else
#define STAMP(i)
#define COUNT(i, v)
#endif
  for (int u0 = 0; u0 < count;) {
    // T quarter waves per bidder: with few bidders left a bidder's superblocks are dealt out to 2 or 4 quarters
    const int rem = count - u0;
    constexpr int kQuarters = kBidWaves * 4;  // quarter waves of the workgroup (qd below runs over them)
    const int tsh = rem <= kQuarters / 4 ? 2 : (rem <= kQuarters / 2 ? 1 : 0), T = 1 << tsh;  // uniform in the workgroup
    const int qd = wave * 4 + row;
    const int u = u0 + (qd >> tsh), part = qd & (T - 1);
    const bool active = u < count;  // uniform within the quarter
    int jj = 0, rank = 0, ba = -1, bb = -1;  // ba, bb: the blocks of the two previous favourites
    float x1 = 0.f, y1 = 0.f, z1 = 0.f, cm = -1e9f;
    float cthr = 3.0e38f, base = 0.f;  // the reach of a box = coarse_threshold(cm, base, the box's A'max): box_within_priced
    if (active) {
      // two round trips: {coordinates, previous favourites} of the bidder, then the favourites' coordinates and
      // prices BY INDEX (the caller's array and the price array of the award phase; their stream positions, needed
      // for the lists only, arrive beside them)
      if (u < kStash) {  // left there by the compaction
        jj = c.stash[u].tgt;
        rank = c.stash[u].rank;
      } else {
        const int2 jr = ldc2(&lst[2 * u]);  // {bidder index, Morton rank}
        jj = jr.x;
        rank = jr.y;
      }
      x1 = c.p1[jj * 3 + 0];
      y1 = c.p1[jj * 3 + 1];
      z1 = c.p1[jj * 3 + 2];
      const int pa = ldc(&c.A.bid[c.o + jj]), pb = ldc(&c.A.bid2[c.o + jj]);
      if (pa >= 0 && pb >= 0) {
        const float *ta = c.p2 + pa * 3, *tb = c.p2 + pb * 3;
        const float da = bid_value(ta[0], ta[1], ta[2], ldc(&c.price[pa]), x1, y1, z1);
        const float db = bid_value(tb[0], tb[1], tb[2], ldc(&c.price[pb]), x1, y1, z1);
        cm = __builtin_fminf(da, db);
        ba = c.rk2[pa] >> 4;
        bb = c.rk2[pb] >> 4;
      }
      {
#pragma clang fp contract(off)
        const float xx = (x1 * x1 + y1 * y1) + z1 * z1;
        base = 2.f * 3.814697265625e-06f * (c.tmax + xx);
        cthr = filter_thr(cm);
      }
    }
    Top2 top = {-1e9f, -1e9f, -1, -1};
    float qlb = -1e9f;  // the team's second-best value after its first round (see below)
    bool first = true;
    STAMP(0)
    // groups of 16 blocks within reach: lane c tests groups c, 16 + c, 32 + c, 48 + c
    unsigned long long hm = 0;
#pragma unroll
    for (int h0 = 0; h0 < kScanBlk / 16; h0 += 16) {  // nh <= 64: the four turns' box reads go out together
      const int h = h0 + col < nh ? h0 + col : 0;
      const bool w = h0 + col < nh && box_within_priced(SL.hb[h][0], SL.hb[h][1], x1, y1, z1, cthr, base);
      hm |= ((__ballot(w) >> (16 * row)) & 0xffffull) << h0;
    }
    auto fetch = [&](ScanCand &B, int i, int cnt) {
      // the round's four list entries in ONE 8-byte LDS read; entries past the quarter's count are stale or
      // uninitialised words: clamped to a block of this cloud, loaded, and ignored by process()
      const uint2 e4 = *reinterpret_cast<const uint2 *>(lq + i);
      const unsigned id[4] = {e4.x & 0xffffu, e4.x >> 16, e4.y & 0xffffu, e4.y >> 16};
#pragma unroll
      for (int q = 0; q < kRoundC; ++q) {
        const int pos = 16 * (int)(id[q] < (unsigned)nblk ? id[q] : 0u) + col;
        B.t[q] = t4[pos];
        B.p[q] = ldc(reinterpret_cast<const float *>(pkc + pos));
      }
    };
    auto process = [&](const ScanCand &B, int i, int cnt) {
      COUNT(6, 1)
      const float lb = __builtin_fmaxf(top.better, qlb);
      const float cthr = filter_thr(__builtin_fmaxf(cm, lb));
      float sqv[kRoundC];
      unsigned mask = 0;
#pragma unroll
      for (int e = 0; e < kRoundC; ++e) {
        sqv[e] = sq_dist(B.t[e].x, B.t[e].y, B.t[e].z, x1, y1, z1);
        const bool pass = i + e < cnt && filter_pass(sqv[e], filter_target(B.p[e]), cthr);
        mask |= pass ? 1u << e : 0u;
      }
      while (__any(mask != 0u)) {  // what passes (rare) takes the reference's arithmetic, one candidate per lane and turn
        COUNT(12, __popcll(__ballot(mask != 0u)))
        COUNT(13, 1)
        if (mask != 0u) {
          const int e = __builtin_ctz(mask);
          mask &= mask - 1u;
          float sq = sqv[0], pr = B.p[0], kf = B.t[0].w;
#pragma unroll
          for (int q = 1; q < kRoundC; ++q) {
            sq = e == q ? sqv[q] : sq;
            pr = e == q ? B.p[q] : pr;
            kf = e == q ? B.t[q].w : kf;
          }
          const float d = (float)((3.0 - (double)__builtin_sqrtf(sq)) - (double)pr);
          top2_push(top, d, __float_as_int(kf), geom);
        }
      }
    };
    // the lanes (and the quarters that share the bidder) hold disjoint targets: the second largest value any of them
    // has seen bounds the final `better` from below
    auto share = [&]() {
      float m1 = top.best, m2 = top.better;
#define SN_SHARE_STEP(o1_, o2_)                                                  \
      {                                                                          \
        const float o1 = o1_, o2 = o2_;                                          \
        m2 = __builtin_fmaxf(__builtin_fminf(m1, o1), __builtin_fmaxf(m2, o2)); \
        m1 = __builtin_fmaxf(m1, o1);                                            \
      }
      SN_SHARE_STEP(SN_DPP_F(m1, 0xB1), SN_DPP_F(m2, 0xB1))
      SN_SHARE_STEP(SN_DPP_F(m1, 0x4E), SN_DPP_F(m2, 0x4E))
      SN_SHARE_STEP(SN_DPP_F(m1, 0x141), SN_DPP_F(m2, 0x141))
      SN_SHARE_STEP(SN_DPP_F(m1, 0x140), SN_DPP_F(m2, 0x140))
      for (int d = 16; d < 16 * T; d <<= 1) SN_SHARE_STEP(__shfl_xor(m1, d), __shfl_xor(m2, d))
#undef SN_SHARE_STEP
      qlb = __builtin_fmaxf(qlb, m2);
    };
#if SN_SCAN_RESHARE > 0
    int rounds = 0;
#endif
    do {
      // The blocks within reach into the lists of the team's quarters, dealt out in turn.  The blocks of the two
      // previous favourites go first: after the round that holds them the team knows two values near the final
      // top two, shares the second one once (`qlb`), and hardly anything passes the filter afterwards.
      int all = 0, cnt = 0;
      if (first && ba >= 0) {
        if (part == 0) lq[0] = (unsigned short)ba;
        all = 1;
        if (bb != ba) {
          if ((1 & (T - 1)) == part) lq[1 >> tsh] = (unsigned short)bb;
          all = 2;
        }
        cnt = (all + T - 1 - part) >> tsh;
      }
      // (the quarters of a team must stop at the same group: the test is on `all`, which they share, not on `cnt`)
      while (__any(hm != 0ull && ((all + T - 1) >> tsh) <= kScanList - 32)) {  // two groups per turn: their box reads overlap
        if (hm != 0ull && ((all + T - 1) >> tsh) <= kScanList - 32) {           // uniform within the TEAM
          const int h0 = __builtin_ctzll(hm);
          hm &= hm - 1ull;
          const bool two = hm != 0ull;
          const int h1 = two ? __builtin_ctzll(hm) : h0;
          hm = two ? hm & (hm - 1ull) : hm;
          const int s0 = 16 * h0 + col, s1 = 16 * h1 + col;
          const f4 a0 = SL.blk[s0][0], c0 = SL.blk[s0][1], a1 = SL.blk[s1][0], c1 = SL.blk[s1][1];
          const bool w0 = s0 != ba && s0 != bb && box_within_priced(a0, c0, x1, y1, z1, cthr, base);
          const bool w1 = two && s1 != ba && s1 != bb && box_within_priced(a1, c1, x1, y1, z1, cthr, base);
          const unsigned b0 = (unsigned)((__ballot(w0) >> (16 * row)) & 0xffffull);
          const unsigned b1 = (unsigned)((__ballot(w1) >> (16 * row)) & 0xffffull);
          const unsigned below = (1u << col) - 1u;
          const int p0 = all + __popc(b0 & below);
          if (w0 && (p0 & (T - 1)) == part) lq[p0 >> tsh] = (unsigned short)s0;
          all += __popc(b0);
          const int p1 = all + __popc(b1 & below);
          if (w1 && (p1 & (T - 1)) == part) lq[p1 >> tsh] = (unsigned short)s1;
          all += __popc(b1);
          cnt = (all + T - 1 - part) >> tsh;
        }
      }
      asm volatile("" ::: "memory");
      STAMP(1)
      COUNT(8, __builtin_amdgcn_readfirstlane(cnt))
      // their targets, round by round, the next round's loads in flight during a round
      if (__any(cnt > 0)) {
        ScanCand A, B;
        fetch(A, 0, cnt);
        for (int i = 0;;) {
          const bool more = __any(i + kRoundC < cnt);
          if (more) fetch(B, i + kRoundC, cnt);
          process(A, i, cnt);
          if (first) {  // uniform: the first round of the bidder
            first = false;
            share();
          }
#if SN_SCAN_RESHARE > 0
          else if ((++rounds % SN_SCAN_RESHARE) == 0) share();
#endif
          i += kRoundC;
          if (!more) break;
          const bool more2 = __any(i + kRoundC < cnt);
          if (more2) fetch(A, i + kRoundC, cnt);
          process(B, i, cnt);
#if SN_SCAN_RESHARE > 0
          if ((++rounds % SN_SCAN_RESHARE) == 0) share();
#endif
          i += kRoundC;
          if (!more2) break;
        }
      }
      first = false;
      asm volatile("" ::: "memory");
#ifndef SN_SCAN_NO_RETIGHTEN
      // (Only on such data -- c.offsurf: on uniform clouds and on a prediction that lies on the targets' surface the
      // first reach is nearly the final one and the extra exchange cost 2 % of the call.)
      // A second SWEEP for the heaviest bidders (a wave instead of a quarter each, through a table in LDS) was
      // measured on top: 3.87 -> 3.70 ms on the scattered probe at 32 clouds with the threshold at 8 groups, the early
      // iterations -- where many bidders are "heavy" and serialise in the table -- slower, uniform clouds +4 %
      // (the barrier that completes the table); not kept.
      // More groups to go: the blocks listed next are tested against the reach of what the team has SEEN by now, not
      // of what it knew at the start (the two previous favourites at today's prices, or iteration 0's seeds).  The
      // precise filter's bound max(cm, lane's better, qlb) is never below max(cm, qlb), so a block outside this reach
      // holds nothing that passes it.  What it buys depends on the data: on a prediction scattered around the
      // targets 1 % of the bidders start with 200-380 blocks within reach (tools/sim, iteration 0) and were the
      // reason a workgroup took three times as long as the average one; after the first list (<= 48 blocks, the
      // favourites' first) the bound is close to final and most of the rest fails the box test.
      if (c.offsurf && __any(hm != 0ull)) {
#pragma clang fp contract(off)
        share();
        cthr = active ? filter_thr(__builtin_fmaxf(cm, qlb)) : cthr;
      }
#endif
      STAMP(2)
    } while (__any(hm != 0ull));
    // the quarter's 16 partial results: xor 1, xor 2, mirror within 8, mirror within 16; then the team's quarters
#define SN_TOP2_STEP(ctrl)                                                                              \
    {                                                                                                   \
      const float ob = SN_DPP_F(top.best, ctrl), obb = SN_DPP_F(top.better, ctrl);                      \
      const int oi = SN_DPP_I(top.best_i, ctrl), oi2 = SN_DPP_I(top.better_i, ctrl);                    \
      top2_merge(top, ob, obb, oi, oi2, geom);                                                          \
    }
    SN_TOP2_STEP(0xB1)
    SN_TOP2_STEP(0x4E)
    SN_TOP2_STEP(0x141)
    SN_TOP2_STEP(0x140)
#undef SN_TOP2_STEP
    for (int d = 16; d < 16 * T; d <<= 1) {  // uniform
      const float ob = __shfl_xor(top.best, d), obb = __shfl_xor(top.better, d);
      const int oi = __shfl_xor(top.best_i, d), oi2 = __shfl_xor(top.better_i, d);
      top2_merge(top, ob, obb, oi, oi2, geom);
    }
    if (active && col == 0 && part == 0) emit_bid(c.A, c.o, jj, rank, u, c.stash, top, c.eps);
    STAMP(4)
    u0 += kQuarters >> tsh;
  }
#ifdef SN_BID_STAMPS
  if (c.stamps && lane == 0)
    for (int i = 0; i < 16; ++i) c.stamps[i] += st[i];
#endif
#undef STAMP
#undef COUNT
}

struct AuctionArgs {
  int B, n, iters;
  float eps;
  const float *xyz1, *xyz2;
  int *assignment;
  float *dist;
  EmdWs ws;
  AuctionCtl *ctl;
  unsigned *sticky;   // host-visible per-device error word (nullptr: none)
  unsigned spin_limit;
  int safe;           // 1: fenced barriers + agent-scope stores only (SN_EMD_SAFE / failed self-test)
  long long *stats;
  TeamGeom tg;
  int diag;  // SN_EMD_DIAG (tools/emd_ab.py): dwords[4..11] += 100 MHz ticks of team 0 / workgroup 0 per phase
             // (compact, -, bid, barrier, award, barrier, -, -); 2: also per iteration and
             // workgroup at dwords[16 + ((it * 8 + m) * 8 + phase)].  dwords = the diag area of the workspace
             // (sn_emd_diag_offset), zeroed by the call.  Bit 2 (4): every team a mixed-XCD one; bit 3 (8): the
             // second workgroup of team 0 leaves at once and barriers give up early (tests the time-out path).
  long long *dwords;
  int diag_m0;   // SN_EMD_DIAG_M0: the first of the eight workgroups of team 0 whose per-iteration times are recorded
  int scan_max;  // iterations with at most this many bidders in the workgroup take bid_scan (0: never)
  int prof;      // 1: record the execution window in the control block (sn_prof_enable)
  int skip_mode;     // contested-auction handling (outbid-skip + scan everywhere): 0 never, 1 when the lists say so (default), 2 always
  int spread_mode;   // interleaved rank split in scan iterations: 0 never, 1 default, 2 in every iteration
};

// The kernel's LDS, carved from the DYNAMIC segment on purpose: with a static 101 KB the compiler derives "one
// workgroup per CU = 4 waves per SIMD" from the LDS size and hands every wave 128 VGPRs whatever the occupancy
// attributes say, which leaves no register for anybody else on the CU.  With SN_EMD_OCC waves per SIMD asked for
// (5 -> 96 VGPRs) a quarter of every SIMD's register file stays free, and workgroups of OTHER launches (the
// renderer's gather: 52 VGPRs, no LDS) run beside the auction in the issue slots its waves leave idle while they
// wait (wait_frac 0.66).
struct AuctionLds {
  WaveTab tabs[kBidWaves];
  GroupAcc gacc[kBidWaves];
  BidStash stash[kStash];
  int wsum[kBidWaves];
  int s_flag, s_ticket, s_stray, s_range[4], s_bins[kRankBins];
  int s_long, s_skipped;
  ScanLds scan;
};

// (Round 4 also built this body a second time for 96 VGPRs per wave -- amdgpu_waves_per_eu(5, 5), a quarter of every
// SIMD's registers left to co-resident launches -- selectable per call: with the quarter-wave scan the 96-register
// build is 17 % slower ALONE (2.46 against 2.10 ms at 32 clouds, 1.48 against 1.16 at 4: the scan's two candidate
// rounds in flight spill) and the step with the renderer beside it lost 0.2-0.4 ms; removed)
__device__ __forceinline__ void auction_body(const AuctionArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char auction_lds[];
  AuctionLds &L = *reinterpret_cast<AuctionLds *>(auction_lds);
  WaveTab *tabs = L.tabs;
  GroupAcc *gacc = L.gacc;
  BidStash *stash = L.stash;
  int *wsum = L.wsum;
  int &s_flag = L.s_flag, &s_ticket = L.s_ticket, &s_stray = L.s_stray;
  int *s_range = L.s_range, *s_bins = L.s_bins;
  if (threadIdx.x == 0) {
    L.s_long = 0;
    L.s_skipped = 0;
  }
#ifdef SN_EMD_PRIO
  __builtin_amdgcn_s_setprio(SN_EMD_PRIO);  // the chain of dependent steps goes first; co-resident waves fill the gaps
#endif
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
    s_flag = __hip_atomic_load(&a.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;  // a late starter
  }
  __syncthreads();
  const bool late = s_flag != 0;  // one read for the whole workgroup: the branch below must be uniform
  const int ticket = s_ticket;
  if (ticket < 0) return;  // no slot left: surplus workgroup of a grid larger than teams x G
  const int team = ticket / G, m = ticket % G;
  if (team >= a.tg.teams || team >= a.B) return;  // a team without a cloud (fewer clouds than teams)
  if (a.prof && tid == 0)
    atomicMax(&a.ctl->t_first_inv, ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
  if ((a.diag & 8) && ticket == 1) return;        // test knob: a team member that never arrives
  const int n = a.n, nsb = n >> 6;
  TeamSync ts = {a.ctl->bar + (size_t)team * 32, &a.ctl->abort, a.sticky, 0u, a.spin_limit, G, a.safe};
  // A barrier gave up (a team member never arrived within the spin limit: the device is shared with something that
  // keeps a CU from this launch, or a debugger): every workgroup of the launch leaves.  What it leaves behind must
  // not look like a result: the clouds this team had not finished get NaN distances and -1 assignments, the
  // device's sticky word makes the next sn_emd_* call fail (the reference returns an error code there,
  // emd_cuda.cu:276-281).
  auto bail = [&](int b_from) {
    for (int b = b_from; b < a.B; b += a.tg.teams)
      for (int e = tid; e < n; e += kBidThreads) {
        a.dist[(size_t)b * n + e] = __builtin_nanf("");
        a.assignment[(size_t)b * n + e] = -1;
      }
  };
  if (late) {  // the launch was given up before this workgroup started
    bail(team);
    return;
  }
  // Is the whole team on ONE XCD?  Then its stores may stay in that XCD's L2 (see stc).  Every member that took
  // a slot of another XCD's team says so in the team's second control word; one formation barrier later every
  // member reads the same answer.
  bool loc = false;
  if (a.tg.xcd) {
    unsigned *mixed = a.ctl->bar + (size_t)team * 32 + 1;
    if (G > 1) {
      if (tid == 0 && s_stray) __hip_atomic_fetch_or(mixed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!team_barrier(ts, &s_flag)) {
        bail(team);
        return;
      }
      loc = __hip_atomic_load(mixed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    } else {
      loc = true;  // a team of one workgroup
    }
  }
  if (a.safe) loc = false;

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int Rs = n / G, rs0 = m * Rs;  // static slice (final distances); n % 1024 == 0, G <= 64
  const int binsize = n / kRankBins;   // ranks per counter bin (a multiple of 4: n % 1024 == 0)
  // the transposed bin order of scan iterations (see the split): G = 2^gsh workgroups, K = 256 / G bins each,
  // position p = bin ((p mod K) << gsh) + (p >> ksh); only when a bin is a power-of-two number of 4-rank words
  // (the shifts are derived where they are used: four more values live across the bid phase cost five spills)
  auto transposed_bin = [&](int p) {
    const int gsh = 31 - __builtin_clz((unsigned)G);
    return ((p & ((kRankBins >> gsh) - 1)) << gsh) + (p >> (8 - gsh));
  };
  const int block_cnt = n / 1024;
  float *price = a.ws.price;
  int *flags = a.ws.flags;
  int *llist_all = a.ws.list[0];
  BidOut bo = {a.ws.bid, a.ws.bid2, a.ws.rec, a.ws.max_inc, a.ws.head, loc, 0};
  unsigned *const note = a.ctl->bar + (size_t)(a.tg.teams + team) * 32 + kNoteWord;  // cont[2] of this team (see kNoteWord)
  int cloud_seq = 0;
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
    c.p2 = a.xyz2 + o * 3;
    c.price = a.ws.price + o;
    c.t4 = a.ws.t4s + o;
    c.pkc = a.ws.pk + o;
    c.rk2 = a.ws.rank2 + o;
    c.ms = a.ws.mstream + (size_t)b * nsb * 64;
    c.sbb = a.ws.sbbox + (size_t)b * nsb * 32;
    c.A = bo;
    c.stash = stash;
    // stamps grow from cloud to cloud and from iteration to iteration: a word left by an earlier cloud or iteration
    // never looks like this iteration's
    const unsigned stamp0 = (unsigned)cloud_seq * (unsigned)(a.iters + 1);
    ++cloud_seq;
    // the prediction lies OFF the targets (more than a quarter of the bidders farther from their nearest seed than a
    // quarter of that seed's block is wide; counted by emd_seed_kernel): see the bid phase
    const bool far = a.ws.far != nullptr && 4 * a.ws.far[b] > n;
    // bid_scan's copy of the cloud's box hierarchy (constant for the whole call)
    const bool scan_ok = a.scan_max > 0 && 4 * nsb <= kScanBlk;  // nsb % 16 == 0 (n % 1024 == 0)
    if (scan_ok) {
      __syncthreads();  // a team that serves several clouds: nobody still reads the previous cloud's boxes
      {
        const f4 *src = reinterpret_cast<const f4 *>(c.sbb);  // [n / 16] rows {lo x, lo y, lo z, hi x}, {hi y, hi z, -, -}
        f4 *dst = &L.scan.blk[0][0];
        for (int i = tid; i < 8 * nsb; i += kBidThreads) dst[i] = src[i];
      }
      __syncthreads();
      for (int h = tid; h < (nsb >> 2); h += kBidThreads) {
        f4 lo = L.scan.blk[16 * h][0], hi = L.scan.blk[16 * h][1];
        for (int q = 1; q < 16; ++q) {
          const f4 l2 = L.scan.blk[16 * h + q][0], h2 = L.scan.blk[16 * h + q][1];
          lo.x = __builtin_fminf(lo.x, l2.x);
          lo.y = __builtin_fminf(lo.y, l2.y);
          lo.z = __builtin_fminf(lo.z, l2.z);
          lo.w = __builtin_fmaxf(lo.w, l2.w);
          hi.x = __builtin_fmaxf(hi.x, h2.x);
          hi.y = __builtin_fmaxf(hi.y, h2.y);
        }
        L.scan.hb[h][0] = lo;
        L.scan.hb[h][1] = hi;
      }
      __syncthreads();
      // the boxes' price bounds (the free third word of a row's second half; see box_within_priced): the price floor's
      // for a start -- prices only rise for eps >= 0, so it stays valid until an iteration refreshes it
      {
        const float amax0 = filter_target(0.f) + 9.5367431640625e-07f;
        for (int i = tid; i < 4 * nsb; i += kBidThreads) L.scan.blk[i][1].z = amax0;
        for (int h = tid; h < (nsb >> 2); h += kBidThreads) L.scan.hb[h][1].z = amax0;
      }
      __syncthreads();
    }

    int bmin_it = -1000;  // the iteration of the last refresh of the boxes' price bounds (see the bid phase)
    for (int it = 0; it < a.iters; ++it) {
      const int cur = it & 1;
      // (Round 4 measured a deterministic COST-weighted split here -- every bin weighted with the work bid_scan's
      // bidders of that bin needed in the previous iteration (groups of 256 targets within reach), so that skewed data
      // would not leave a team waiting for the workgroup with the expensive bidders: on the scattered-prediction
      // probe the wait at the first barrier went from 12.9 to 11.5 us per late iteration and the call from 2.59 to
      // 2.62 ms, on uniform clouds the call got 4-7 % slower (2.03 -> 2.12 ms at 32 clouds: 12 more spilled
      // registers in a kernel that sits at its 128-VGPR limit).  The imbalance there is in the exact evaluations of
      // a few bidders with most of the cloud within reach, which a per-bin average does not predict.  Not kept.)
      // The unassigned bidders per 1/256 of the rank range, counted by the previous award phase: every workgroup
      // of the team derives the same split of the ranks into G contiguous, equally loaded ranges (bins are
      // indivisible).  A static split left the team waiting ~20 us per late iteration for the workgroup
      // whose region happened to hold two groups of bidders instead of one.
      //
      // WHICH bins a workgroup owns (round 5).  A contiguous range of the Hilbert ranks keeps a workgroup's bidders
      // spatial neighbours -- what the matrix-core search needs (a group of 64 list neighbours shares its reach) --
      // but what a bidder COSTS is a property of where it lies: on a prediction scattered around the targets' surface
      // a bidder's reach holds 25 blocks on average and up to 330 (tools/sim/auction_regime_stats.c), the expensive
      // ones are neighbours, and the team waited at the first barrier of every iteration for the workgroup that owned
      // them (1.25 of a 2.6 ms call at 4 clouds, 16.5 of 79 ms on the refine stages of an untrained generator at 32:
      // profiles/r05_a_emd_regimes_before.txt).  In the iterations that bid with the SCAN (one bidder per quarter
      // wave: nothing shared between list neighbours) the same count-balanced split therefore runs over the bins in
      // TRANSPOSED order, position p = bin (p mod K) G + p div K, K = 256 / G: a workgroup's run of positions is a
      // comb of bins spread evenly over the whole curve, every workgroup a sample of every region.  No protocol, no
      // atomics; the bidders' order inside a list never enters a result.
      const unsigned stamp = stamp0 + (unsigned)it + 1u;
      const bool last = it == a.iters - 1;
      if (wave == 0) {  // lane l holds the bins (positions) 4 l .. 4 l + 3
        const int2 lo2 = ldc2(a.ws.bins[cur] + b * kRankBins + 4 * lane);
        const int2 hi2 = ldc2(a.ws.bins[cur] + b * kRankBins + 4 * lane + 2);
        int contw = 0;
        if (lane == 0) contw = ldc(reinterpret_cast<const int *>(&note[cur]));
        contw = __builtin_amdgcn_readfirstlane(contw);
        int v[4] = {lo2.x, lo2.y, hi2.x, hi2.y};
        int lsum = (v[0] + v[1]) + (v[2] + v[3]);
        int incl = lsum;
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d);
          if (lane >= d) incl += t;
        }
        const int total = __shfl(incl, 63);
        // contested auction (emit_bid's outbid-skip, the scan for every workgroup): for eps >= 0, not in the last
        // iteration, when the previous iteration's award phase said so
        const bool contested = a.eps >= 0.f && !last && (a.skip_mode == 2 || (a.skip_mode == 1 && (unsigned)contw == stamp));
        const int smax = ((contested || far) && a.scan_max > 0 && a.scan_max < kScanContested) ? kScanContested : a.scan_max;
        // (only where the cost per bidder is what varies: off-surface / contested data.  On uniform clouds and on a
        // prediction on the targets' surface the transposed order costs 1.5-4 % -- a workgroup's bidders no longer
        // share target blocks in its L1 -- and balances nothing: SN_EMD_SPREAD=3 spreads every scan iteration.)
        const bool spread_ok = G > 1 && G <= kRankBins / 4 && ((binsize >> 2) & ((binsize >> 2) - 1)) == 0;
        const bool spread = spread_ok && scan_ok &&
                            (a.spread_mode == 2 || ((a.spread_mode == 3 || (a.spread_mode == 1 && (contested || far))) &&
                                                    4 * total <= 3 * G * smax));
#ifdef SN_EMD_NOSPREAD
        if (false) {
#else
        if (spread) {  // the same counters in transposed order (through LDS: a wave's own writes, in order)
#endif
#pragma unroll
          for (int q = 0; q < 4; ++q) s_bins[4 * lane + q] = v[q];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int p = 4 * lane + q;
            v[q] = s_bins[transposed_bin(p)];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) s_bins[4 * lane + q] = 0;  // the award phase counts in these
          lsum = (v[0] + v[1]) + (v[2] + v[3]);
          incl = lsum;
          for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
          }
        }
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
          s_range[1] = first;
          s_range[2] = cnt;
          s_range[3] = (contested ? 1 : 0) | (spread ? 2 : 0);
        }
      }
      __syncthreads();
      const int U = s_range[0], p0 = s_range[1], R = s_range[2] * binsize;
      if (U == 0) break;  // every workgroup of the team reads the same value
      const bool contested = (s_range[3] & 1) != 0, spread = (s_range[3] & 2) != 0;
      const int r0 = p0 * binsize;  // where the workgroup's list lives (by position: disjoint whatever the order)
      int *llist = llist_all + 2 * (o + r0);   // {index, rank} pairs
      if (m == 0 && tid == 0 && a.stats) {
        atomicAdd(reinterpret_cast<unsigned long long *>(a.stats), (unsigned long long)U * n);
        if (b == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.stats) + 1, 1ULL);
      }
      // the counters the award phase fills in this iteration (read last at the top of the previous one)
      if (m == 0 && tid >= 64 && tid < 64 + kRankBins)
        stc(loc, a.ws.bins[cur ^ 1] + b * kRankBins + (tid - 64), 0);
      const bool dg = a.diag && team == 0 && tid == 0;
      long long tk = dg ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
      auto tick = [&](int slot) {
        if (dg) {
          const long long now = (long long)__builtin_amdgcn_s_memrealtime();
          if (m == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.dwords) + slot, (unsigned long long)(now - tk));
          if (a.diag >= 2 && m >= a.diag_m0 && m < a.diag_m0 + 8 && it < 64)
            a.dwords[16 + ((it * 8 + (m - a.diag_m0)) * 8 + (slot - 4))] = now - tk;
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
          // word w of the workgroup's positions: position p0 + w / wpb, i.e. bin p (contiguous) or its transpose
          int rw = r0 + 4 * w;
          if (w < vec) {  // coherent reads (other workgroups raised these flags), two 8-byte words per lane
#ifdef SN_EMD_NOSPREAD
            if (false) {
#else
            if (spread) {
#endif
              const int wsh = 31 - __builtin_clz((unsigned)(binsize >> 2));
              rw = transposed_bin(p0 + (w >> wsh)) * binsize + 4 * (w & ((1 << wsh) - 1));
            }
            const int2 lo2 = ldc2(flags + o + rw), hi2 = ldc2(flags + o + rw + 2);
            f = make_int4(lo2.x, lo2.y, hi2.x, hi2.y);
            if (f.x | f.y) stc2(loc, flags + o + rw, 0, 0);
            if (f.z | f.w) stc2(loc, flags + o + rw + 2, 0, 0);
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
            const int r = rw;
            // the list entry also waits in the slot's (still unused) stash for bid_scan: one round trip less
            auto put = [&](int j, int rr) {
              stc2(loc, &llist[2 * pos], j, rr);
              if (pos < kStash) {
                stash[pos].tgt = j;
                stash[pos].rank = rr;
              }
              ++pos;
            };
            if (f.x) put(f.x - 1, r);
            if (f.y) put(f.y - 1, r + 1);
            if (f.z) put(f.z - 1, r + 2);
            if (f.w) put(f.w - 1, r + 3);
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
#ifdef SN_BID_STAMPS
#ifndef SN_STAMP_FROM   // experiment builds: the iterations whose bid phase is stamped (default: the tail, 10 .. iters - 1)
#define SN_STAMP_FROM 10
#define SN_STAMP_TO 1000000
#endif
        c.stamps = (a.diag && team == 0 && m < 3 && it >= SN_STAMP_FROM && it <= SN_STAMP_TO) ? a.dwords + 16 + 3200 + (m * 16 + wave) * 16 : nullptr;
#endif
        // A CONTESTED auction (the team's `cont` word, read at the top: the same answer in every workgroup) or a
        // prediction that lies OFF the targets (`far`, from the seed kernel): hundreds of far bidders per near target
        // are the data on which the matrix-core search finds most pairs "near" and pays for every hit in its queue,
        // so every workgroup bids with the scan -- the refine stages of an untrained generator at 32 clouds: 56.7 ->
        // 29.7 ms per call, uniform clouds 2.45 -> 2.57 (profiles/r05_a_emd_handoff_not_kept.txt, SN_EMD_SCAN=2048) --
        // and a contested iteration's bidders check the target's maximum before they link themselves (emit_bid).
        c.A.skip = contested ? 1 : 0;
        c.offsurf = contested || far;
        const int scan_max = ((contested || far) && a.scan_max > 0 && a.scan_max < kScanContested) ? kScanContested : a.scan_max;
        if (scan_ok && Um <= scan_max) {  // uniform in the workgroup: a quarter wave per bidder
#ifndef SN_EMD_NO_BMIN
          if (contested && a.eps >= 0.f && it - bmin_it >= SN_EMD_BMIN_EVERY) {
            // CONTESTED clouds: every box's bound from the SMALLEST PRICE it holds now (prices do not move during a bid
            // phase).  A bidder there sits far from the surface, the near-side targets' prices have climbed, and with the
            // price floor's bound their blocks -- which it can no longer want -- stay within its reach.  The refresh
            // costs 11-15 us per workgroup (16 coherent loads per thread + two barriers): in every iteration of every
            // off-surface cloud it made the SCATTERED regime slower (3.9 -> 4.45 ms per call at 32 clouds, its prices stay
            // low) while the untrained regime gained (31.2 -> 26.7): hence contested iterations only, and every fourth
            // one -- a bound from earlier, lower prices stays valid (eps >= 0: prices only rise).
            bmin_it = it;
            for (int bq = tid; bq < 4 * nsb; bq += kBidThreads) {
              const float2 *pp = c.pkc + 16 * bq;
              float pm = 3.0e38f;
#pragma unroll 8
              for (int i = 0; i < 16; ++i) pm = __builtin_fminf(pm, ldc_pk(pp + i).x);
              L.scan.blk[bq][1].z = filter_target(pm) + 9.5367431640625e-07f;
            }
            __syncthreads();
            for (int h = tid; h < (nsb >> 2); h += kBidThreads) {
              float mx = L.scan.blk[16 * h][1].z;
              for (int q = 1; q < 16; ++q) mx = __builtin_fmaxf(mx, L.scan.blk[16 * h + q][1].z);
              L.scan.hb[h][1].z = mx;
            }
            __syncthreads();
          } else
#endif
          if (a.eps < 0.f) {  // prices may fall: this iteration's price floor for every box
            for (int i = tid; i < 4 * nsb; i += kBidThreads) L.scan.blk[i][1].z = c.a_max;
            for (int h = tid; h < (nsb >> 2); h += kBidThreads) L.scan.hb[h][1].z = c.a_max;
            __syncthreads();
          }
          bid_scan(c, L.scan, llist, Um, wave, lane);
        } else {
          const int ngroups = (Um + 63) >> 6;
          int S = 1;
          while (S < kBidWaves && S * 2 * ngroups <= kBidWaves) S *= 2;
          const int gpb = kBidWaves / S;
          const int seg = wave & (S - 1), gslot = wave / S;
          for (int q0 = 0; q0 < ngroups; q0 += gpb) {
            bid_group(c, tabs[wave], gacc[gslot], llist, Um, q0 + gslot, ngroups, S, seg, lane);
            if (q0 + gpb < ngroups) __syncthreads();
          }
        }
      }
      if (a.diag) __syncthreads();
      tick(6);
      if (!team_barrier(ts, &s_flag)) {
        bail(b);
        return;
      }
      tick(7);
      // ---- award: GetMax (emd_cuda.cu:181-194) and Assign (:196-215) in one target-centric pass.
      // Every bidder linked itself into its target's list during the bid phase; the bidder that is still the
      // list's head after the barrier walks it.  With mi = the target's max_increments (the atomic maximum of
      // this iteration's increments and whatever an earlier iteration left there, exactly the reference's
      // tensor), the bidders inside the window |inc - mi| <= 1e-6 (double compare, :188) compete, the HIGHEST
      // bidder index wins (the sequential order of :188-191); with nobody inside the window -- only possible
      // for eps < 0, when every increment is below the tensor's initial 0 -- the persistent max_idx entry of an
      // earlier iteration decides, as in the reference, which never clears that tensor (:181-194,
      // emd_module.py:50).  The winner gets the target (eviction, price update, :203-211), every other bidder
      // of the list is flagged to bid again.  In the last iteration every bidder takes its target (:200).
      // One walker per target: no word is written by two threads, and what a walker reads was written before
      // the barrier (lists, increments) or belongs to its target alone.
      {
        unsigned *nextbins = reinterpret_cast<unsigned *>(a.ws.bins[cur ^ 1] + b * kRankBins);
        auto raise = [&](int rank, int j) {  // the flag is the bidder's index + 1; counted per bin in LDS first
          stc(loc, &flags[o + rank], j + 1);
          atomicAdd(&s_bins[rank / binsize], 1);
        };
        const int *rec = a.ws.rec + 4 * o;
        for (int u = tid; u < Um; u += kBidThreads) {
          int tgt, rank, inc_bits, nxt, jown;
          if (u < kStash) {
            const BidStash sb = stash[u];
            tgt = sb.tgt;
            rank = sb.rank;
            inc_bits = sb.inc_bits;
            nxt = sb.next;
            jown = sb.j;
          } else {
            const int2 jr = ldc2(&llist[2 * u]);
            tgt = ldc(&bo.bid[o + jr.x]);
            rank = jr.y;
            jown = jr.x;
            const int2 r2 = ldc2(&rec[4 * rank]);
            inc_bits = r2.x;
            nxt = r2.y;
          }
          if (tgt < 0) {  // no bid (non-finite input): stays unassigned, distance 0, zero gradient
            if (!last) raise(rank, jown);
            continue;
          }
          if (nxt == kOutbid) {  // outbid on arrival, never linked (emit_bid; not in the last iteration)
            raise(rank, jown);
            atomicAdd(&L.s_skipped, 1);
            continue;
          }
          // the list's head and the four words of the target, requested TOGETHER: most targets have one bidder, which
          // is then the head and needs them -- one round trip instead of two on the award phase's chain
          const int hd = ldc(&bo.head[o + tgt]);
          const float mi = ldc(&bo.max_inc[o + tgt]);
          const int wp = ldc(&a.ws.max_idx[o + tgt]);
          const int inv = ldc(&a.ws.assignment_inv[o + tgt]);
          const float pr = ldc(&price[o + tgt]);
          if (hd != rank) continue;  // somebody else walks this target's list
          int w_rank = -1, w_j = -1, p_inc = 0, p_j = -1;
          float w_inc = 0.f;
          bool persist_hit = false;
          int steps = 0;
          for (int cr = rank, ci = inc_bits, cn = nxt, cj = jown;; ++steps) {
            const float bi = __int_as_float(ci);
            if (last) {  // forced assignment (:200): every bidder takes its target; prices still accumulate (:209)
              stc(loc, &a.assignment[o + cj], tgt);
              w_inc += bi;  // several claimants: a race in the reference, list order here
            } else if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6) {
              if (w_rank < 0) {
                w_rank = cr;
                w_j = cj;
                w_inc = bi;
              } else if (cj > w_j) {  // several bidders inside the window: the highest bidder INDEX wins
                raise(w_rank, w_j);
                w_rank = cr;
                w_j = cj;
                w_inc = bi;
              } else {
                raise(cr, cj);
              }
            } else if (cr == wp) {  // outside the window, but the persistent winner: decided after the walk
              persist_hit = true;
              p_inc = ci;
              p_j = cj;
            } else {
              raise(cr, cj);
            }
            if (cn < 0) break;
            const int2 r2 = ldc2(&rec[4 * cn]);  // {increment, next} and the index: one line segment, one round trip
            cj = ldc(&rec[4 * cn + 2]);
            cr = cn;
            ci = r2.x;
            cn = r2.y;
          }
          if (steps >= 12) L.s_long = 1;  // one thread followed a dozen dependent loads: see emit_bid's outbid-skip
          stc(loc, &bo.head[o + tgt], -1);
          if (last) {
            stc(loc, &price[o + tgt], pr + w_inc);
            continue;
          }
          if (w_rank >= 0) {
            stc(loc, &a.ws.max_idx[o + tgt], w_rank);
            if (persist_hit) raise(wp, p_j);
          } else if (persist_hit) {
            w_rank = wp;
            w_j = p_j;
            w_inc = __int_as_float(p_inc);
          }
          if (w_rank < 0) continue;  // nobody wins this target in this iteration
          if (inv != -1) {
            stc(loc, &a.assignment[o + inv], -1);
            raise(a.ws.rank1[o + inv], inv);  // evicted: bids again
          }
          stc(loc, &a.ws.assignment_inv[o + tgt], w_j);
          stc(loc, &a.assignment[o + w_j], tgt);
          const float np = pr + w_inc;
          stc(loc, &price[o + tgt], np);
          const int pos = a.ws.rank2[o + tgt];  // the bid phase reads the price by stream position
          stc(loc, reinterpret_cast<float *>(a.ws.pk + o + pos), np);
          stc(loc, a.ws.prt + o + ((pos >> 6) * 64 + (pos & 15) * 4 + ((pos >> 4) & 3)), np);  // [sb][c][q]
          stc(loc, &bo.max_inc[o + tgt], -1e9f);
        }
        __syncthreads();
        if (tid < kRankBins) {
          const int v = s_bins[tid];
          if (v) {
            __hip_atomic_fetch_add(nextbins + tid, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_bins[tid] = 0;
          }
        }
        if (tid == kRankBins) {  // contested targets: the next iteration's bidders check the maximum before they link
          // switched ON by a long list, kept on while at least half of this workgroup's bidders find themselves
          // outbid on arrival (with the skip on the lists are short, so the walkers no longer see the contention);
          // uniform clouds: 20-30 % in the early iterations -- there the second round trip per bid costs 1.6 us per
          // iteration and buys nothing
          if (L.s_long || (L.s_skipped >= 16 && 2 * L.s_skipped >= Um))
            stc(loc, reinterpret_cast<int *>(&note[cur ^ 1]), (int)(stamp + 1u));
          L.s_long = 0;
          L.s_skipped = 0;
        }
      }
      if (a.diag) __syncthreads();
      tick(8);
      if (!team_barrier(ts, &s_flag)) {
        bail(b);
        return;
      }
      tick(9);
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
  if (a.prof && tid == 0) atomicMax(&a.ctl->t_last, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

#ifndef SN_EMD_WAVES_PER_EU
#define SN_EMD_WAVES_PER_EU 4   // 128 VGPRs per wave: 16 waves of one workgroup fill a CU's register files
#endif
__global__ __attribute__((amdgpu_flat_work_group_size(kBidThreads, kBidThreads),
                          amdgpu_waves_per_eu(SN_EMD_WAVES_PER_EU, SN_EMD_WAVES_PER_EU))) void emd_auction_kernel(AuctionArgs a) {
  auction_body(a);
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
constexpr size_t kCtlWords = 32 + 2 * 32 * 1024;                    // ticket, abort, up to 1024 team counters + 1024 note blocks
constexpr size_t kCtlBytes = 4 * kCtlWords + 8 * kDiagWords;

EmdWs carve(void *workspace, int b, int n) {
  char *p = static_cast<char *>(workspace);
  const size_t arr = sn::align_up((size_t)b * n * 4, 256);
  const size_t arr2 = sn::align_up((size_t)b * n * 8, 256);
  EmdWs ws;
  ws.assignment_inv = reinterpret_cast<int *>(p); p += arr;
  ws.price = reinterpret_cast<float *>(p); p += arr;
  ws.bid = reinterpret_cast<int *>(p); p += arr;
  ws.bid2 = reinterpret_cast<int *>(p); p += arr;
  ws.rec = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * n * 16, 256);
  ws.max_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_idx = reinterpret_cast<int *>(p); p += arr;
  ws.head = reinterpret_cast<int *>(p); p += arr;
  ws.list[0] = reinterpret_cast<int *>(p); p += arr2;
  ws.prt = reinterpret_cast<float *>(p); p += arr;
  ws.bins[0] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * kRankBins * 4, 256);
  ws.bins[1] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * kRankBins * 4, 256);
  ws.t4s = reinterpret_cast<f4 *>(p); p += sn::align_up((size_t)b * n * 16, 256);
  ws.pk = reinterpret_cast<float2 *>(p); p += arr2;
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
  ws.far = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * 4, 256);
  ws.ctl = p; p += kCtlBytes;
  return ws;
}

// ---- once per device: does this GPU behave the way the fence-free barriers and the XCD-local plain stores assume?
// The persistent auction hands data between workgroups with relaxed accesses only (see team_barrier, stc).  That is
// gfx950 cache behaviour, not a promise of the HIP memory model, and the XCD of a workgroup is read from a raw
// hardware register.  emd_litmus_kernel replays both hand-overs on every workgroup of a full grid for `rounds`
// rounds: each workgroup publishes a word per round (a) with an agent-scope (sc1) store and (b) with a plain
// (workgroup-scope) store, all workgroups meet at a fence-free barrier, then every workgroup reads every other
// workgroup's (a) word -- and the (b) word of those that reported the same XCC id -- with coherent loads and
// counts stale values.  res[0] = stale agent-scope words, res[1] = stale same-XCD plain words, res[2] = barrier
// time-outs, res[3] = workgroups that ran, res[4 + x] = workgroups that reported XCC id x.
__global__ __launch_bounds__(64) void emd_litmus_kernel(unsigned *ctl, unsigned *xcc_of, unsigned *agent_words,
                                                        unsigned *plain_words, unsigned *res, int W, int rounds) {
  __shared__ int s_me, s_ok;
  const int lane = threadIdx.x;
  if (lane == 0) {
    s_me = (int)__hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ok = 1;
  }
  __syncthreads();
  const int me = s_me;
  if (me >= W) return;
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
  if (lane == 0) {
    __hip_atomic_store(&xcc_of[me], xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(&res[3], 1u);
    atomicAdd(&res[4 + xcc], 1u);
  }
  unsigned target = 0;
  auto barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += (unsigned)W;
    if (lane == 0) {
      __hip_atomic_fetch_add(&ctl[32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(&ctl[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 21) || __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_ok = 0;
          break;
        }
      }
    }
    __syncthreads();
    return s_ok != 0;
  };
  if (!barrier()) {
    if (lane == 0) atomicAdd(&res[2], 1u);
    return;
  }
  unsigned bad_agent = 0, bad_plain = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (lane == 0) {
      // a different line every round (a line that is already shared would hide a missing write-through)
      __hip_atomic_store(&agent_words[(size_t)(r & 7) * W * 16 + (size_t)me * 16], (unsigned)(r * 65536 + me),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&plain_words[(size_t)(r & 7) * W * 16 + (size_t)me * 16], (unsigned)(r * 65536 + me),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (!barrier()) {
      if (lane == 0) atomicAdd(&res[2], 1u);
      return;
    }
    for (int v = lane; v < W; v += 64) {
      const unsigned want = (unsigned)(r * 65536 + v);
      const unsigned ga = __hip_atomic_load(&agent_words[(size_t)(r & 7) * W * 16 + (size_t)v * 16], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
      bad_agent += ga != want;
      if (__hip_atomic_load(&xcc_of[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc) {
        const unsigned gp = __hip_atomic_load(&plain_words[(size_t)(r & 7) * W * 16 + (size_t)v * 16],
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad_plain += gp != want;
      }
    }
    if (!barrier()) {  // nobody overwrites a slot (8 rounds later) before everybody has read it
      if (lane == 0) atomicAdd(&res[2], 1u);
      return;
    }
  }
  if (bad_agent) atomicAdd(&res[0], bad_agent);
  if (bad_plain) atomicAdd(&res[1], bad_plain);
}

struct DeviceState {
  int verified = 0;       // 0: not yet, 1: fence-free + XCD-local paths verified, 2: fall back (fenced, agent-scope stores)
  int tries = 0;
  int cus = 0;            // compute units the verdict was reached with: a different count (a compute-partition mode
                          // change between calls gives the same ordinal another shape) voids it
  double next_try = 0.0;  // earliest time (steady clock, seconds) of the next attempt after an undecided one
  char why[160] = {0};
};
std::mutex g_dev_mu;
DeviceState g_dev[64];

double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Runs the litmus on `dev`; fills st.verified / st.why.  Everything goes through a PRIVATE non-blocking stream
// (asynchronous memset, launch and copy, then a wait for that stream only): no null-stream operation, no
// device-wide synchronisation, nothing that other streams of the process order themselves against.  The caller has
// checked that ITS stream is not being captured; if some other thread holds a global-mode capture, hipMalloc fails
// and the test stays undecided (the call at hand takes the fenced path, a later call tries again).  Undecided runs
// (the grid was not co-resident: a busy device) never latch the slow path: the next attempt is allowed a little
// later, with a growing pause (0.25 s ... 8 s), so a device that is busy at start-up still gets verified.
void verify_device(DeviceState &st, int dev, int cus) {
  st.tries++;
  st.cus = cus;
  const double pause = 0.25 * (double)(1 << (st.tries < 6 ? st.tries - 1 : 5));
  st.next_try = now_seconds() + pause;
  const int W = cus, rounds = 24;
  const size_t words = 64 + (size_t)W + 2 * 8 * (size_t)W * 16 + 16;
  unsigned *buf = nullptr;
  hipStream_t ps = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&buf), words * 4) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(st.why, sizeof st.why, "self-test: hipMalloc failed (undecided, will retry)");
    return;
  }
  if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(buf);
    snprintf(st.why, sizeof st.why, "self-test: no private stream (undecided, will retry)");
    return;
  }
  unsigned *ctl = buf, *xcc_of = buf + 64, *aw = xcc_of + W, *pw = aw + 8 * (size_t)W * 16, *res = pw + 8 * (size_t)W * 16;
  unsigned r[16] = {0};
  hipError_t e = hipMemsetAsync(buf, 0, words * 4, ps);
  if (e == hipSuccess) {
    emd_litmus_kernel<<<W, 64, 0, ps>>>(ctl, xcc_of, aw, pw, res, W, rounds);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(r, res, sizeof r, hipMemcpyDeviceToHost, ps);
  if (e == hipSuccess) e = hipStreamSynchronize(ps);
  (void)hipStreamDestroy(ps);
  (void)hipFree(buf);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    st.verified = 2;
    snprintf(st.why, sizeof st.why, "self-test: %s", hipGetErrorString(e));
    return;
  }
  if (r[2] != 0 || (int)r[3] != W) {  // the grid was not co-resident (busy device): undecided, try again later
    snprintf(st.why, sizeof st.why, "self-test undecided after %d tries (device busy: %u of %d workgroups ran, %u time-outs)",
             st.tries, r[3], W, r[2]);
    return;
  }
  if (r[0] != 0 || r[1] != 0) {
    st.verified = 2;
    snprintf(st.why, sizeof st.why, "self-test: %u stale agent-scope words, %u stale same-XCD plain words", r[0], r[1]);
    return;
  }
  st.verified = 1;
  st.why[0] = 0;
}

// The launch's own execution window, added to a per-device accumulator {ticks, launches} by a one-thread launch behind
// the auction (measurement aid: only with sn_prof_enable).  HIP events around the launch also contain its wait for
// compute units -- the persistent grid needs every CU empty, and in bench.py's auction-first order the previous step's
// renderer is still draining when the launch is enqueued -- which says something about the schedule, not the kernel.
__global__ void emd_exec_window_kernel(const AuctionCtl *ctl, unsigned long long *acc) {
  const unsigned long long first = ~ctl->t_first_inv, last = ctl->t_last;
  if (ctl->t_first_inv != 0ull && last > first) {
    acc[0] += last - first;
    acc[1] += 1ull;
  }
}
std::mutex g_exec_mu;
unsigned long long *g_exec_acc[64] = {nullptr};
unsigned long long *exec_accumulator(int dev) {  // nullptr if it cannot be allocated (then nothing is recorded)
  std::lock_guard<std::mutex> lk(g_exec_mu);
  if (!g_exec_acc[dev]) {
    void *p = nullptr;
    if (hipMalloc(&p, 16) == hipSuccess && hipMemset(p, 0, 16) == hipSuccess) g_exec_acc[dev] = static_cast<unsigned long long *>(p);
    else (void)hipGetLastError();
  }
  return g_exec_acc[dev];
}

}  // namespace

// measurement aid next to sn_prof_read("emd_auction"): launches recorded since the last reset on the current device and
// the sum of their execution windows (first working workgroup's start to the last one's end, in-kernel 100 MHz clock)
extern "C" long long sn_emd_prof_exec(double *total_ms, int reset) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  unsigned long long *acc = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_exec_mu);
    acc = g_exec_acc[dev];
  }
  unsigned long long v[2] = {0, 0};
  if (acc) {
    if (hipMemcpy(v, acc, 16, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
    if (reset) (void)hipMemset(acc, 0, 16);
  }
  if (total_ms) *total_ms = (double)v[0] * 1e-5;
  return (long long)v[1];
}

extern "C" size_t sn_emd_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  return 14 * sn::align_up((size_t)b * n * 4, 256) + 2 * sn::align_up((size_t)b * n * 8, 256) +
         2 * sn::align_up((size_t)b * kRankBins * 4, 256) + 3 * sn::align_up((size_t)b * n * 16, 256) +
         2 * (size_t)b * kSortCells * 4 + 2 * sn::align_up((size_t)b * 24, 256) +
         sn::align_up((size_t)b * (n / 16) * 32, 256) + sn::align_up((size_t)b * 4, 256) + kCtlBytes;
}

// byte offset of the diagnostic words (SN_EMD_DIAG) inside the workspace
extern "C" size_t sn_emd_diag_offset(int b, int n) {
  return sn_emd_workspace_bytes(b, n) - 8 * kDiagWords;
}

// 0: the fence-free / XCD-local paths are in use on the current device; 2: the library fell back to fenced
// barriers and agent-scope stores (SN_EMD_SAFE=1 or a failed self-test: sn_last_error() says why); -1: not decided
// yet (no sn_emd_forward call on this device so far).
extern "C" int sn_emd_mode(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  const char *e = SN_KNOB("SN_EMD_SAFE");
  if (e && e[0] == '1') return 2;
  if (g_dev[dev].verified == 2) {
    sn::fail(0, "%s", g_dev[dev].why);
    return 2;
  }
  return g_dev[dev].verified == 1 ? 0 : -1;
}

// Explicit start-up self-test of the current device (what the first sn_emd_forward otherwise does lazily): runs the
// litmus now and returns sn_emd_mode()'s answer (0 / 2), or -1 when the run was undecided (busy device; call again).
// Call it from the thread that owns the device, outside any graph capture.
extern "C" int sn_emd_selftest(void) {
  int dev = 0, cus = 0;
  SN_HIP(hipGetDevice(&dev));
  SN_REQUIRE(dev >= 0 && dev < 64, "sn_emd_selftest: unexpected device ordinal %d", dev);
  SN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    DeviceState &st = g_dev[dev];
    if (st.verified != 0 && st.cus != cus) st = DeviceState{};
    if (st.verified == 0) verify_device(st, dev, cus);
    (void)hipGetLastError();
  }
  return sn_emd_mode();
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
  SN_REFUSE_CAPTURE(s, "sn_emd_forward");
  int dev = 0, cus = 0;
  SN_HIP(hipGetDevice(&dev));
  SN_REQUIRE(dev >= 0 && dev < 64, "sn_emd_forward: unexpected device ordinal %d", dev);
  if (const int rc = sn::check_sticky(dev, "sn_emd_forward")) return rc;
  SN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  SN_REQUIRE(cus >= 1 && cus <= 1024, "sn_emd_forward: unexpected compute-unit count %d", cus);
  const char *safe_env = SN_KNOB("SN_EMD_SAFE");   // once per process; per call under SN_KNOBS_PER_CALL=1 (tests)
  int safe = safe_env && safe_env[0] == '1';
  unsigned *sticky = sn::sticky_device_word(dev);
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    DeviceState &st = g_dev[dev];
    if (st.verified != 0 && st.cus != cus) st = DeviceState{};   // the device changed shape: verify again
    if (!safe && st.verified == 0 && now_seconds() >= st.next_try) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(s, &cap);
      if (cap == hipStreamCaptureStatusNone) {
        verify_device(st, dev, cus);
        if (st.verified == 2) fprintf(stderr, "sparenet_hip: EMD falls back to fenced barriers on device %d: %s\n", dev, st.why);
      }
      (void)hipGetLastError();
    }
    if (st.verified != 1) safe = 1;   // unverified (yet) or failed: the conservative path
  }
  const EmdWs ws = carve(workspace, b, n);
  const long total = (long)b * n;
  const int eblocks = (int)((total + kThreads - 1) / kThreads < 2048 ? (total + kThreads - 1) / kThreads : 2048);
  // both clouds' Hilbert sorts in one launch; the bidders' cell ids go to `flags` for the moment (every word of it
  // is written by emd_init_kernel afterwards)
  SN_REQUIRE(cloud_sort_pair(b, SortSide{n, xyz2, ws.bbox, ws.hist, ws.cell_of, ws.tperm},
                             SortSide{n, xyz1, ws.bbox1, ws.hist1, ws.flags, ws.perm1}, s) == 0,
             "sn_emd_forward: cannot size the sort kernel's LDS");
  static const bool check = [] { const char *e = getenv("SN_EMD_CHECK"); return e && e[0] == '1'; }();
  emd_init_kernel<<<eblocks, kThreads, 0, s>>>(b, n, xyz2, assignment, ws);
  emd_sbbox_kernel<<<(int)(((long)b * (n / 64) + 3) / 4), 256, 0, s>>>(b, n, xyz2, ws);
  emd_seed_kernel<<<eblocks, kThreads, 0, s>>>(b, n, xyz1, xyz2, ws);
  {
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
    args.sticky = sticky;
    args.safe = safe;
    args.stats = stats;
    static const int diag = [] { const char *e = getenv("SN_EMD_DIAG"); return e ? atoi(e) : 0; }();
    static const int gmax = [] { const char *e = getenv("SN_EMD_G"); const int v = e ? atoi(e) : 64; return v >= 1 ? v : 64; }();
    static const int legacy = [] { const char *e = getenv("SN_EMD_GEOM"); return e && e[0] == '1' ? 1 : 0; }();
    {  // once per process; per call under SN_KNOBS_PER_CALL=1 (the tests compare the two bid paths inside one process)
      const char *e = SN_KNOB("SN_EMD_SCAN");
      const int v = e ? atoi(e) : SN_EMD_SCAN_MAX;
      args.scan_max = v < 0 ? 0 : v;
    }
    args.tg = team_geometry(b, cus * kWgPerCu, gmax, legacy);
    args.diag = diag;
    static const int diag_m0 = [] { const char *e = getenv("SN_EMD_DIAG_M0"); return e ? atoi(e) : 0; }();
    args.diag_m0 = diag_m0;
    static const unsigned spin_env = [] { const char *e = getenv("SN_EMD_SPIN_LIMIT"); return e ? (unsigned)atol(e) : 0u; }();
    args.spin_limit = (diag & 8) ? (1u << 15) : (spin_env ? spin_env : kSpinLimit);   // SN_EMD_SPIN_LIMIT: debugging aid
    args.dwords = reinterpret_cast<long long *>(static_cast<char *>(ws.ctl) + 4 * kCtlWords);
    {  // once per process; per call under SN_KNOBS_PER_CALL=1 (the tests compare the settings inside one process)
      const char *e = SN_KNOB("SN_EMD_SKIP");
      args.skip_mode = e ? atoi(e) : 1;
      e = SN_KNOB("SN_EMD_SPREAD");
      args.spread_mode = e ? atoi(e) : 1;
    }
    SN_REQUIRE(args.tg.teams <= 1024, "sn_emd_forward: too many teams (%d)", args.tg.teams);
    if (diag) SN_HIP(hipMemsetAsync(args.dwords, 0, 8 * kDiagWords, s));
    SN_HIP(hipMemsetAsync(ws.ctl, 0, 4 * (32 + 2 * 32 * (size_t)args.tg.teams), s));   // barrier blocks + note blocks
    {
      // 140 KB of dynamic LDS: above the 64 KB a launch may ask for unannounced.  Every call: the attribute belongs
      // to the CURRENT device (several devices per process), and a failure must be reported by the call that meets it.
      const hipError_t lds_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&emd_auction_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AuctionLds));
      SN_REQUIRE(lds_rc == hipSuccess, "sn_emd_forward: hipFuncSetAttribute(%zu bytes of LDS): %s", sizeof(AuctionLds),
                 hipGetErrorString(lds_rc));
      unsigned long long *exec_acc = nullptr;
      args.prof = 0;
      if (sn::prof_enabled() && !sn::capturing(s)) {
        exec_acc = exec_accumulator(dev);
        args.prof = exec_acc != nullptr;
      }
      sn::PersistentLaunch chain(dev, s);  // never beside another team-waiting launch of this process (common.hpp)
      SN_TIMED("emd_auction", s, (emd_auction_kernel<<<cus * kWgPerCu, kBidThreads, sizeof(AuctionLds), s>>>(args)));
      if (args.prof) emd_exec_window_kernel<<<1, 1, 0, s>>>(args.ctl, exec_acc);
    }
    if (check) {  // debugging aid: wait for the launch and report a time-out at once
      unsigned abort_word = 0;
      SN_HIP(hipStreamSynchronize(s));
      SN_HIP(hipMemcpy(&abort_word, &args.ctl->abort, 4, hipMemcpyDeviceToHost));
      if (abort_word != 0) sn::clear_sticky(dev);
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
  {
    int dev = 0;
    SN_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64)
      if (const int rc = sn::check_sticky(dev, "sn_emd_backward")) return rc;
  }
  const long total = (long)b * n;
  const int blocks = (int)((total + kThreads - 1) / kThreads < 2048 ? (total + kThreads - 1) / kThreads : 2048);
  emd_bwd_kernel<<<blocks, kThreads, 0, sn::as_stream(stream)>>>(b, n, xyz1, xyz2, graddist,
                                                                 assignment, gradxyz1);
  return sn::launch_status("sn_emd_backward");
}
