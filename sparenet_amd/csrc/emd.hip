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
//   * 4 launches per iteration instead of 7: the unassigned list for the next iteration is
//     produced by Assign itself (losers and evicted points append with a wave-aggregated
//     atomic) -- no count / scan / compaction kernels.
//   * Bid is the hot kernel (fp32 VALU bound).  Every lane of a wave is a different bidder
//     and the wave walks a wave-UNIFORM target stream that arrives through the scalar cache
//     in SGPRs (one s_load_dwordx16 = 4 targets) -- no LDS, no barrier in the scan.
//   * fp32 packed filter + exact path: a target is evaluated exactly (correctly rounded
//     sqrt, fp64 detour, top-2 update) only if a conservative fp32 test says it could enter
//     the bidder's top-2; thresholds are seeded from the bidder's previous two favourites.
//     ~2-4 % of the wave-steps take the exact path; results stay bit-identical.
//   * the unassigned count is only known on the device, so a fixed XCD-aware grid adapts:
//     S = 2^k <= 64 waves share one group of 64 bidders, each scanning n/S targets; up to
//     16 of them merge in LDS, the rest through emd_bid_finish_kernel.
//   * GetMax: deterministic atomicMax of the bidder index inside the window.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;     // element-wise kernels
constexpr int kBlocksPerCloud = 32;  // bid kernel: 32 workgroups x 16 waves = 512 waves per cloud (swept 8..64)

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

typedef float f2 __attribute__((ext_vector_type(2)));

// FILTER-ONLY squared distance of two targets at once: fused multiply-adds (6 packed ops
// instead of 8).  It differs from the exact (dx*dx + dy*dy) + dz*dz by <= 2 ulp, which the
// filter margins absorb; every target that passes is re-evaluated exactly by sq_dist.
__device__ __forceinline__ f2 sq_dist2_fast(f2 tx, f2 ty, f2 tz, float x1, float y1, float z1) {
  const f2 dx = tx - x1, dy = ty - y1, dz = tz - z1;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
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
    atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

// Prepared target stream: one 32-byte record per PAIR of targets,
//   [x0 x1 | y0 y1 | z0 z1 | A'0 A'1],  A'_k = filter_target(price_k),
// so that a wave-uniform s_load_dwordx16 delivers 4 targets straight into SGPR pairs that
// the packed fp32 ops consume as operands.  Written by emd_init_kernel, A' refreshed by
// emd_assign_kernel for the targets whose price changed.
__device__ __forceinline__ size_t tgt_slot(int k, int field) {
  return (size_t)(k >> 1) * 8 + field * 2 + (k & 1);
}

struct EmdWs {
  int *assignment_inv;
  float *price;
  int *bid, *bid2;
  float *bid_inc;
  float *max_inc;
  int *max_idx;
  int *list[2];
  int *cnt[2];
  float *tgt;  // [B, n/2, 8] prepared target stream (see tgt_slot)
  float *partial;  // [B][16][4][64] float4
};

__global__ void emd_init_kernel(int B, int n, const float *__restrict__ xyz2,
                                int *__restrict__ assignment, EmdWs ws) {
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
    ws.list[0][e] = (int)(e % n);
    {
      const long bb = e / n;
      const int k = (int)(e - bb * n);
      float *t = ws.tgt + bb * n * 4;
      t[tgt_slot(k, 0)] = xyz2[e * 3 + 0];
      t[tgt_slot(k, 1)] = xyz2[e * 3 + 1];
      t[tgt_slot(k, 2)] = xyz2[e * 3 + 2];
      t[tgt_slot(k, 3)] = filter_target(0.f);
    }
    if (e < B) {
      ws.cnt[0][e] = n;
      ws.cnt[1][e] = 0;
    }
  }
}

struct BidOut {
  int *bid, *bid2;
  float *bid_inc, *max_inc;
  int *max_idx;
  float4 *partial;  // [B][kMaxSplitGroups][kMaxSplit][64] cross-workgroup partial top-2's
};

constexpr int kBidWaves = 16;
constexpr int kBidThreads = kBidWaves * 64;
constexpr int kMaxSegments = 64;                          // waves per bidder group, at most
constexpr int kMaxSplit = kMaxSegments / kBidWaves;       // workgroups per bidder group, at most
constexpr int kMaxSplitGroups = 16;                       // groups per cloud that may be split

// How a cloud's G*16 waves are spread over its ngroups groups of 64 bidders:
// S_total = 2^k <= 64 waves per group (each scanning n/S_total targets); up to 16 of them live
// in one workgroup (LDS merge), the rest in sibling workgroups (merge in emd_bid_finish_kernel).
struct BidSplit {
  int s_total, s_block, nb;
};
__host__ __device__ inline BidSplit bid_split(int ngroups, int G) {
  int s = 1;
  while (s < kMaxSegments && s * 2 * ngroups <= G * kBidWaves) s *= 2;
  if (s > kBidWaves && ngroups > kMaxSplitGroups) s = kBidWaves;
  BidSplit r;
  r.s_total = s;
  r.s_block = s < kBidWaves ? s : kBidWaves;
  r.nb = s / r.s_block;
  return r;
}

__device__ __forceinline__ void emit_bid(const BidOut &A, size_t o, int j, const Top2 &top,
                                         float eps) {
  const float inc = (top.best - top.better) + eps;
  A.bid[o + j] = top.best_i;
  A.bid2[o + j] = top.better_i == top.best_i ? -1 : top.better_i;
  A.bid_inc[o + j] = inc;
  atomic_max_float(&A.max_inc[o + top.best_i], inc);
  A.max_idx[o + top.best_i] = -1;  // winner is re-derived by emd_getmax_kernel
}

// ---------------------------------------------------------------------------------------
// Bid kernel.  Every lane of a wave is a DIFFERENT bidder and the whole wave walks the SAME
// targets, so targets are wave-uniform and travel through the scalar cache into SGPRs
// (measured: feeding them through LDS as broadcast ds_read_b128 is LDS-issue bound at
// ~45 cycles per target step; the packed-math filter itself needs ~26).
// A workgroup has 16 waves.  S = 2^k <= 16 waves share one group of 64 bidders, wave s
// scanning targets [s n/S, (s+1) n/S); the S partial top-2's meet in LDS.  S is chosen on
// the device from the unassigned count so that the fixed grid stays busy when few bidders
// are left (tail iterations), and is 1 while there are >= 64 * waves bidders.
// ---------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const float cfloat;
typedef __attribute__((address_space(4))) const f4 cf4;

// constant address space (4): nothing in the bid kernel writes the target stream or the
// prices, and a uniform load from AS4 is always selected as a scalar (SMEM) load; the 64-bit
// base goes through readfirstlane so that it provably lives in SGPRs.
__device__ __forceinline__ const cfloat *uniform_ptr(const float *q) {
  const unsigned long long a = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const cfloat *)(((unsigned long long)hi << 32) | lo);
}

// 8 waves/SIMD (<= 64 VGPRs): every wave has at most two scalar loads in flight, so the
// scalar-cache latency is hidden by wave-level parallelism
__global__ __launch_bounds__(kBidThreads, 8) void emd_bid_kernel(
    int B, int G, int n, float eps, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ price, const float *__restrict__ tgt,
    const int *__restrict__ list, const int *__restrict__ cnt, BidOut A,
    long long *__restrict__ stats) {
  __shared__ float m_best[kBidWaves][64], m_better[kBidWaves][64];
  __shared__ int m_bi[kBidWaves][64], m_bi2[kBidWaves][64];
  // XCD-aware decode of the 1-D grid: workgroup `lin` runs on XCD lin % 8 and every cloud's
  // workgroups share that residue, so a cloud's 256 KB target stream + prices stay in ONE
  // 4 MB L2 (4 clouds per XCD at B = 32) instead of all 8 MB cycling through every L2.
  const int lin = blockIdx.x;
  const int xcd = lin & 7, rr = lin >> 3;
  const int b = (rr / G) * 8 + xcd;
  const int bx = rr % G;  // this workgroup's index among its cloud's G workgroups
  if (b >= B) return;
  const int U = cnt[b];
  if (U == 0) return;
  if (stats && bx == 0 && threadIdx.x == 0) {
    atomicAdd(reinterpret_cast<unsigned long long *>(stats), (unsigned long long)U * n);
    if (b == 0) atomicAdd(reinterpret_cast<unsigned long long *>(stats) + 1, 1ULL);
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // provably uniform
  const int lane = threadIdx.x & 63;
  const size_t o = (size_t)b * n;
  const float *__restrict__ p1 = xyz1 + o * 3;
  const float *__restrict__ p2 = xyz2 + o * 3;
  const float *__restrict__ pr = price + o;
  const int *__restrict__ lst = list + o;
  const cfloat *tg = uniform_ptr(tgt + o * 4);
  const cfloat *prc = uniform_ptr(price + o);

  const int ngroups = (U + 63) >> 6;
  const BidSplit sp = bid_split(ngroups, G);
  const int S = sp.s_block;        // segments (waves) of one group inside this workgroup
  const int gpb = kBidWaves / S;   // bidder groups per workgroup (1 when the group is split)
  const int seg = wave & (S - 1);  // this wave's segment within the workgroup
  const int gslot = wave / S;
  const int seg_len = n / sp.s_total;  // n % 1024 == 0, s_total <= 64: a multiple of 16

  // reference partition, only needed to order exact ties (emd_cuda.cu:108-109,136)
  const int block_cnt = n / 1024;
  const TieGeom geom = {n, 1024 / ((U + block_cnt - 1) / block_cnt)};

  // work items: (group, part) with part < nb; a workgroup takes gpb consecutive groups (nb == 1)
  // or one (group, part) (nb > 1)
  for (int q0 = bx * gpb; q0 < ngroups * sp.nb; q0 += G * gpb) {  // uniform per block
    const int grp = sp.nb > 1 ? q0 / sp.nb : q0 + gslot;
    const int part = sp.nb > 1 ? q0 % sp.nb : 0;
    const int k_begin = (part * S + seg) * seg_len, k_end = k_begin + seg_len;
    const int u = grp * 64 + lane;
    const bool active = grp < ngroups && u < U;
    const int j = lst[active ? u : 0];
    const float x1 = p1[j * 3 + 0], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
    Top2 top = {-1e9f, -1e9f, -1, -1};

    // seed the filter with the bidder's previous two favourites under today's prices:
    // both are real targets, so the final `better` is at least the smaller of the two.
    float cm = -1e9f;
    {
      const int pa = A.bid[o + j], pb = A.bid2[o + j];
      if (pa >= 0 && pb >= 0) {
        const float da = bid_value(p2[pa * 3], p2[pa * 3 + 1], p2[pa * 3 + 2], pr[pa], x1, y1, z1);
        const float db = bid_value(p2[pb * 3], p2[pb * 3 + 1], p2[pb * 3 + 2], pr[pb], x1, y1, z1);
        cm = __builtin_fminf(da, db);
      }
    }
    float cthr = filter_thr(cm);

    if (grp < ngroups) {  // wave-uniform
      // software pipelined: the next 4 targets' record pair (s_load_dwordx16) is in flight
      // while the current 4 are evaluated
      const cf4 *rec = (const cf4 *)(tg + (size_t)(k_begin >> 1) * 8);
      f4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
      for (int k = k_begin; k < k_end; k += 4) {
        const f4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;  // [x0 x1 y0 y1][z0 z1 a0 a1] x 2
        rec += (k + 4 < k_end) ? 4 : 0;
        n0 = rec[0];
        n1 = rec[1];
        n2 = rec[2];
        n3 = rec[3];
        const f2 s01 = sq_dist2_fast(f2{c0.x, c0.y}, f2{c0.z, c0.w}, f2{c1.x, c1.y}, x1, y1, z1);
        const f2 s23 = sq_dist2_fast(f2{c2.x, c2.y}, f2{c2.z, c2.w}, f2{c3.x, c3.y}, x1, y1, z1);
        const f2 r01 = f2{c1.z, c1.w} - cthr, r23 = f2{c3.z, c3.w} - cthr;
        // s <= R |R|: a negative R (target too expensive to matter at any distance) never passes
        const float t0 = r01.x * __builtin_fabsf(r01.x), t1 = r01.y * __builtin_fabsf(r01.y);
        const float t2 = r23.x * __builtin_fabsf(r23.x), t3 = r23.y * __builtin_fabsf(r23.y);
        const bool pass[4] = {s01.x <= t0, s01.y <= t1, s23.x <= t2, s23.y <= t3};
        // one wave-uniform branch per 4 targets; the exact path is out of line.  A filter
        // evaluated with an older (looser) threshold only passes more, never less.
        if (__builtin_expect(__any(pass[0] | pass[1] | pass[2] | pass[3]), 0)) {
          const int ku = __builtin_amdgcn_readfirstlane(k);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (__any(pass[i])) {
              // exact re-evaluation (separately rounded products and sums) from the stream
              const cfloat *rr = tg + (size_t)((ku + i) >> 1) * 8 + ((ku + i) & 1);
              const float sq = sq_dist(rr[0], rr[2], rr[4], x1, y1, z1);
              const float d = (float)((3.0 - (double)__builtin_sqrtf(sq)) - (double)prc[ku + i]);
              if (pass[i]) top2_push(top, d, ku + i, geom);
              cm = __builtin_fmaxf(cm, top.better);
              cthr = filter_thr(cm);
            }
          }
        }
      }
    }
    // merge the S partial results of a bidder group (segment 0's wave collects)
    if (S > 1) {
      m_best[wave][lane] = top.best;
      m_better[wave][lane] = top.better;
      m_bi[wave][lane] = top.best_i;
      m_bi2[wave][lane] = top.better_i;
      __syncthreads();
      if (seg == 0)
        for (int w = wave + 1; w < wave + S; ++w)
          top2_merge(top, m_best[w][lane], m_better[w][lane], m_bi[w][lane], m_bi2[w][lane], geom);
      __syncthreads();  // LDS merge slots are reused by the next group
    }
    if (seg == 0 && grp < ngroups) {
      if (sp.nb == 1) {
        if (active) emit_bid(A, o, j, top, eps);
      } else {  // publish this workgroup's partial; emd_bid_finish_kernel merges the nb of them
        A.partial[(((size_t)b * kMaxSplitGroups + grp) * kMaxSplit + part) * 64 + lane] =
            make_float4(top.best, top.better, __int_as_float(top.best_i), __int_as_float(top.better_i));
      }
    }
  }
}

// merges the partial top-2's of bidder groups that were split over several workgroups
__global__ __launch_bounds__(kBidThreads) void emd_bid_finish_kernel(
    int G, int n, float eps, const int *__restrict__ list, const int *__restrict__ cnt, BidOut A) {
  const int b = blockIdx.x;
  const int U = cnt[b];
  if (U == 0) return;
  const int ngroups = (U + 63) >> 6;
  const BidSplit sp = bid_split(ngroups, G);
  if (sp.nb == 1) return;
  const int grp = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = grp * 64 + lane;
  if (grp >= ngroups || u >= U) return;
  const size_t o = (size_t)b * n;
  const int block_cnt = n / 1024;
  const TieGeom geom = {n, 1024 / ((U + block_cnt - 1) / block_cnt)};
  const float4 *pp = A.partial + ((size_t)b * kMaxSplitGroups + grp) * kMaxSplit * 64 + lane;
  const float4 p0 = pp[0];
  Top2 top = {p0.x, p0.y, __float_as_int(p0.z), __float_as_int(p0.w)};
  for (int q = 1; q < sp.nb; ++q) {
    const float4 pq = pp[(size_t)q * 64];
    top2_merge(top, pq.x, pq.y, __float_as_int(pq.z), __float_as_int(pq.w), geom);
  }
  emit_bid(A, o, list[o + u], top, eps);
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
    float *__restrict__ tgt_stream, int last) {
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
      const float np = price[o + tgt] + bid_inc[o + j];
      price[o + tgt] = np;
      tgt_stream[o * 4 + tgt_slot(tgt, 3)] = filter_target(np);  // keep the bid filter in sync
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
  ws.bid2 = reinterpret_cast<int *>(p); p += arr;
  ws.bid_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_inc = reinterpret_cast<float *>(p); p += arr;
  ws.max_idx = reinterpret_cast<int *>(p); p += arr;
  ws.list[0] = reinterpret_cast<int *>(p); p += arr;
  ws.list[1] = reinterpret_cast<int *>(p); p += arr;
  ws.cnt[0] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * 4, 256);
  ws.cnt[1] = reinterpret_cast<int *>(p); p += sn::align_up((size_t)b * 4, 256);
  ws.tgt = reinterpret_cast<float *>(p); p += sn::align_up((size_t)b * n * 16, 256);
  ws.partial = reinterpret_cast<float *>(p);
  return ws;
}

}  // namespace

extern "C" size_t sn_emd_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  return 9 * sn::align_up((size_t)b * n * 4, 256) + 2 * sn::align_up((size_t)b * 4, 256) +
         sn::align_up((size_t)b * n * 16, 256) + (size_t)b * 16 * 4 * 64 * 16;
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
  emd_init_kernel<<<eblocks, kThreads, 0, s>>>(b, n, xyz2, assignment, ws);
  const int g_env = kBlocksPerCloud;
  const int bid_grid = g_env * 8 * sn::ceil_div(b, 8);
  const dim3 lin_grid(sn::ceil_div(n, kThreads * 4) < 16 ? sn::ceil_div(n, kThreads * 4) : 16, b);
  for (int it = 0; it < iters; ++it) {
    const int c = it & 1;
    const BidOut bo = {ws.bid, ws.bid2, ws.bid_inc, ws.max_inc, ws.max_idx,
                       reinterpret_cast<float4 *>(ws.partial)};
    SN_TIMED("emd_bid", s, (emd_bid_kernel<<<bid_grid, kBidThreads, 0, s>>>(
        b, g_env, n, eps, xyz1, xyz2, ws.price, ws.tgt, ws.list[c], ws.cnt[c], bo, stats)));
    emd_bid_finish_kernel<<<b, kBidThreads, 0, s>>>(g_env, n, eps, ws.list[c], ws.cnt[c], bo);
    emd_getmax_kernel<<<lin_grid, kThreads, 0, s>>>(n, ws.bid, ws.bid_inc, ws.max_inc, ws.max_idx,
                                                    ws.list[c], ws.cnt[c], ws.cnt[c ^ 1]);
    emd_assign_kernel<<<lin_grid, kThreads, 0, s>>>(n, assignment, ws.assignment_inv, ws.price,
                                                    ws.bid, ws.bid_inc, ws.max_inc, ws.max_idx,
                                                    ws.list[c], ws.cnt[c], ws.list[c ^ 1],
                                                    ws.cnt[c ^ 1], ws.tgt, it == iters - 1);
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
