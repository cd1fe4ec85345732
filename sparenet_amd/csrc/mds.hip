// mds.hip -- minimum density sampling + gather for MI355X (gfx950).
//
// Reference: cuda/MDS/MDS_cuda.cu:91-211 (sampling), :29-79 (gather fwd/bwd),
// binding cuda/MDS/MDS.cpp:54-135.  Semantics: oracle/mds.c.  Greedy: every round
// adds exp(-d/t) (x2 for k >= 8192) of the last pick to every point's density and
// picks the arg-min; ties resolve to argmin (bitrev(k mod bs), k), the order the
// reference's reduction tree induces.  The exponential is sn_expf (shared with the
// oracle, include/sn_expf.h).
//
// MI355X design: the op is 16383 DEPENDENT picks per cloud, so the levers are the latency of one
// round and -- since round 6, on the team kernel that serves SpareNet-sized clouds -- several exact
// picks per exchange between the workgroups of a cloud (mds_dense_team_kernel below).  The
// one-workgroup kernels (small clouds, teams that cannot be formed, graph capture):
// One workgroup (bs <= 1024 lanes = 16 waves = one CU) owns a
// cloud and keeps the WHOLE state in registers: lane tid owns points tid, tid+bs, ...
// (coordinates + density, 4 VGPRs per point, 19 points per lane at n = 19384) -- the
// reference re-reads xyz and read-modify-writes `temp` in global memory every round.
// Arg-min = per-lane scan, wave64 xor-butterfly on a packed 64-bit key
// (density bits << 32 | bitrev(tid) << 8 | slot), one LDS hand-off between the 16
// waves with double buffering => ONE barrier per round (reference: 11).
// (float)((double)temp + w) equals the plain fp32 sum for every pair of floats
// (the double sum is exact unless w < ulp(temp)/32, where both round to temp), so
// the accumulation stays in fp32.
#include <cstdlib>

#include "cloud_sort.hpp"
#include "common.hpp"
#include "wave_dpp.hpp"
#include "../../include/sn_expf.h"

// Round 5's two attempts at the surface regime's round -- several picks per round (exact, 3.3 picks per round, 19.0 ms
// against 18.4) and lazy slot summaries (25.6 ms against 18.3) -- were index-exact and slower; they live in the git
// history (commit 255d4b0 and before) and in profiles/r05_c_mds_rounds_not_kept.txt, not in this file.

#ifdef SN_MDS_STAMPS   // experiment builds (tools/build_variant.sh stamps mds.hip -DSN_MDS_STAMPS): where a round of the
                       // dense-regime team kernel spends its time, wave 0 of member 0 of cloud 0, 100 MHz ticks:
                       // [0] update + arg-min in the wave, [1] first workgroup barrier (waiting for the slowest wave),
                       // [2] store + poll of the team's words, [3] minimum + the pick's coordinates (uniform load),
                       // [4] second workgroup barrier + hand-over, [5] rounds
__device__ unsigned long long g_mds_stamps[8];
extern "C" int sn_mds_debug_stamps(unsigned long long *out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mds_stamps), 64) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mds_stamps), z, 64) != hipSuccess) return -1;
  }
  return 0;
}
#define MDS_STAMP(i)                                                                    \
  if (stamping) {                                                                       \
    const long long now_ = (long long)__builtin_amdgcn_s_memrealtime();                 \
    st_acc[i] += now_ - st_tk;                                                          \
    st_tk = now_;                                                                       \
  }
#else
#define MDS_STAMP(i)
#endif

namespace {

__device__ __forceinline__ float lane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m, int width) {
  const unsigned lo = __shfl_xor((unsigned)v, m, width);
  const unsigned hi = __shfl_xor((unsigned)(v >> 32), m, width);
  return ((unsigned long long)hi << 32) | lo;
}

// ZLDS: number of coordinates kept in LDS instead of VGPRs (0: none, 1: z, 2: y and z) --
// 19+ points per lane would otherwise exceed the 128-VGPR budget of a 1024-lane
// workgroup and spill.  A lane only ever reads the LDS words it wrote itself.
// BS: compile-time workgroup size (1024) or 0 = run-time (clouds below 2048 points); with
// BS fixed every per-slot index test folds into an immediate instead of a hoisted VGPR.
template <int PPT, int ZLDS, int BS>
__global__ __launch_bounds__(1024) void mds_kernel(int n, int m, const float *__restrict__ xyz,
                                                   const float *__restrict__ mean_mst_length,
                                                   int *__restrict__ idxs, int lg) {
#pragma clang fp contract(off)
  __shared__ unsigned long long wave_key[2][16];
  extern __shared__ __attribute__((aligned(16))) float zs[];
  const int b = blockIdx.x;
  const int bs = BS ? BS : (int)blockDim.x;
  const int tid = threadIdx.x;
  const float *__restrict__ p = xyz + (size_t)b * n * 3;
  int *__restrict__ out = idxs + (size_t)b * m;
  const float mml = mean_mst_length[b];
  const float t = (float)(5.0 * (double)mml * (double)mml);
  const unsigned rev = lg ? (__brev((unsigned)tid) >> (32 - lg)) : 0u;
  const int wave_width = bs < 64 ? bs : 64;
  const int nwaves = (bs + 63) / 64;
  const int lim = n - tid;  // slot i is a real point iff i*bs < lim

  float px[PPT], py[ZLDS == 2 ? 1 : PPT], pz[ZLDS ? 1 : PPT], tmp[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * bs;
    const int kk = k < n ? k : 0;
    px[i] = p[kk * 3 + 0];
    if (ZLDS == 2) {
      if (k < n) {
        zs[2 * k + 0] = p[kk * 3 + 1];
        zs[2 * k + 1] = p[kk * 3 + 2];
      }
    } else if (ZLDS == 1) {
      py[i] = p[kk * 3 + 1];
      if (k < n) zs[k] = p[kk * 3 + 2];
    } else {
      py[i] = p[kk * 3 + 1];
      pz[i] = p[kk * 3 + 2];
    }
    tmp[i] = 0.f;
  }
  int last = 0;
  if (tid == 0) out[0] = 0;

  for (int j = 1; j < m; ++j) {
    const float x1 = p[last * 3 + 0], y1 = p[last * 3 + 1], z1 = p[last * 3 + 2];
    float bestv = 1e9f;
    int bi = 0;
    const int rel = last - tid;  // slot i holds the last pick iff i*bs == rel
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (i * bs < lim) {
        const int k = tid + i * bs;
        float v = (i * bs == rel) ? 1e9f : tmp[i];
        float yk, zk;
        if (ZLDS == 2) {
          const float2 yz = reinterpret_cast<const float2 *>(zs)[k];
          yk = yz.x;
          zk = yz.y;
        } else {
          yk = py[ZLDS == 2 ? 0 : i];
          zk = ZLDS ? zs[k] : pz[ZLDS ? 0 : i];
        }
        const float dx = px[i] - x1, dy = yk - y1, dz = zk - z1;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float e = sn_expf(-d / t);
        v = v + (k < 8192 ? e : e + e);
        tmp[i] = v;
        if (v < bestv) {
          bestv = v;
          bi = i;
        }
      }
      // keep the per-point temporaries of sn_expf from piling up across slots (spills)
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long key =
        ((unsigned long long)__float_as_uint(bestv) << 32) | (unsigned long long)((rev << 8) | (unsigned)bi);
    for (int s = 1; s < wave_width; s <<= 1) {
      const unsigned long long o = shfl_xor_u64(key, s, wave_width);
      key = o < key ? o : key;
    }
    if (nwaves > 1) {
      const int buf = j & 1;
      if ((tid & 63) == 0) wave_key[buf][tid >> 6] = key;
      __syncthreads();
      const int l = tid & 15;
      key = l < nwaves ? wave_key[buf][l] : ~0ull;
      for (int s = 1; s < 16; s <<= 1) {
        const unsigned long long o = shfl_xor_u64(key, s, 16);
        key = o < key ? o : key;
      }
    }
    const unsigned low = (unsigned)key;
    if ((unsigned)(key >> 32) >= __float_as_uint(1e9f)) {
      last = 0;  // nothing below 1e9: every lane reported (1e9, index 0)
    } else {
      const unsigned wrev = low >> 8;
      const int wtid = lg ? (int)(__brev(wrev) >> (32 - lg)) : 0;
      last = wtid + (int)(low & 0xffu) * bs;
    }
    if (tid == 0) out[j] = last;
  }
}

// ---------------------------------------------------------------------------------------
// Cluster-sorted variant (the production path for 2048 <= n <= 20352, e.g. SpareNet's 19384).
// exp(-d/t) is EXACTLY 0 (sn_expf) once d/t >= 104, i.e. outside a ball of radius
// sqrt(104 t) around the last pick (0.19 for SpareNet's t): there the update is a no-op.
// With the reference's ownership (lane tid owns k = tid, tid+1024, ...) every register slot
// of a wave mixes points from all over the cloud, so nothing can be skipped in SIMD.  Here
// the points are first put in Morton order; slot i of wave w holds the 64 spatially adjacent
// points of cluster i*16+w.  Every round one lane per slot tests the cluster's bounding
// box against the ball (one ballot), and only the 1-3 slots that can receive a non-zero
// update are evaluated; the others keep their densities untouched -- bit-identical results,
// ~4x fewer issued ops per round.  The arg-min is order independent: every candidate carries
// the key (density bits, bitrev(k mod 1024), k) of its ORIGINAL index k.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }

// minimum over each row of 16 lanes, left in every lane of the row (four DPP steps)
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
  v = umin32(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));   // quad_perm 1,0,3,2
  v = umin32(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));   // quad_perm 2,3,0,1
  v = umin32(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));  // row_half_mirror
  v = umin32(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// wave-uniform minimum over the 64 lanes: wave_dpp.hpp (one definition for the sampler, the expansion penalty and
// the renderer)
using sn::wave_min_u32;

// sn_expf (include/sn_expf.h) for arguments x <= 0 (or NaN): the same correctly rounded
// operations in the same order, so it returns the bits sn_expf returns.  The x > 88 clamp is
// unreachable here; the final scaling is one v_ldexp_f32, which rounds y * 2^n once -- exactly
// what the header's two-step multiply does for results in the subnormals.
__device__ __forceinline__ float sn_expf_nonpositive(float x) {
  const float magic = 12582912.0f;
  const float t = __builtin_fmaf(x, 1.44269504088896341f, magic);
  const float fn = t - magic;
  const int n = (int)fn;
  float r = __builtin_fmaf(fn, -0.693359375f, x);
  r = __builtin_fmaf(fn, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = __builtin_fmaf(p, r2, r) + 1.0f;
  return x > -104.0f ? __builtin_ldexpf(y, n) : 0.0f;
}

// -d / t.  With rt = RN(1/t), the product refined twice through the exact residual is the
// correctly rounded quotient (Markstein's theorem; tools/probe/div_probe.hip checks it
// exhaustively for 64 divisors x 5.8e8 dividends on the device).  The caller only takes this
// path when no intermediate can overflow and t is far from the subnormals; quotients that
// underflow have |x| < 2^-26, where sn_expf(x) == 1 whatever the last bits of x are.
__device__ __forceinline__ float neg_div(float d, float t, float rt, bool fast) {
  if (!fast) return -d / t;
  const float n = -d;
  float q = n * rt;
  float r = __builtin_fmaf(-t, q, n);
  q = __builtin_fmaf(r, rt, q);
  r = __builtin_fmaf(-t, q, n);
  return __builtin_fmaf(r, rt, q);
}

template <int PPT>
__global__ __launch_bounds__(1024) void mds_clustered_kernel(
    int n, int m, const float *__restrict__ xyz, const int *__restrict__ perm_all,
    const float *__restrict__ bbox, const float *__restrict__ mean_mst_length,
    int *__restrict__ idxs, float team_ratio) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float yz[];  // [PPT*1024][2], lane private
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float *__restrict__ p = xyz + (size_t)b * n * 3;
  const int *__restrict__ perm = perm_all + (size_t)b * n;
  int *__restrict__ out = idxs + (size_t)b * m;
  const float mml = mean_mst_length[b];
  const float t = (float)(5.0 * (double)mml * (double)mml);
  // every point at squared distance >= cut2 contributes sn_expf(-d/t) == 0 exactly
  const float cut2 = 104.0f * t * 1.0001f;
  const float rt = 1.0f / t;
  float diag2 = 0.f;  // the squared diagonal of the cloud's bounding box bounds every d
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ext = bbox[b * 6 + 3 + a] - bbox[b * 6 + a];
    diag2 += ext * ext;
  }
  const bool fast_div = t >= 0x1p-40f && t <= 0x1p40f && diag2 * rt < 0x1p100f;
  // dense regime (the cut ball covers most of the cloud): done by mds_dense_team_kernel when it was launched
  if (team_ratio > 0.f && cut2 > team_ratio * diag2) return;

  float px[PPT], tmp[PPT];
  unsigned low[PPT];  // (bitrev10(k mod 1024) << 16) | (k << 1) | (k >= 8192), ~0 for padding
  // lane i < PPT: axis-aligned bounding box of slot i
  float blx = 3e38f, bly = 3e38f, blz = 3e38f, bhx = -3e38f, bhy = -3e38f, bhz = -3e38f;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int s = ((i * 16 + wave) << 6) + lane;  // sorted position
    const bool valid = s < n;
    const int k = valid ? perm[s] : 0;
    const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
    px[i] = valid ? x : 0.f;
    yz[2 * (i * 1024 + tid) + 0] = valid ? y : 0.f;
    yz[2 * (i * 1024 + tid) + 1] = valid ? z : 0.f;
    // Padding entries start at 1e9 like a picked point: 1e9f + e == 1e9f for every e <= 2, so
    // they stay there without a validity test and never win the arg-min.
    tmp[i] = valid ? 0.f : 1e9f;
    low[i] = valid ? ((__brev((unsigned)(k & 1023)) >> 22) << 16) | ((unsigned)k << 1) | (k >= 8192 ? 1u : 0u)
                   : 0xffffffffu;
    // bounding box of this wave's cluster i (kept by lane i)
    float lx = valid ? x : 3e38f, ly = valid ? y : 3e38f, lz = valid ? z : 3e38f;
    float hx = valid ? x : -3e38f, hy = valid ? y : -3e38f, hz = valid ? z : -3e38f;
    for (int mm = 1; mm < 64; mm <<= 1) {
      lx = __builtin_fminf(lx, __shfl_xor(lx, mm));
      ly = __builtin_fminf(ly, __shfl_xor(ly, mm));
      lz = __builtin_fminf(lz, __shfl_xor(lz, mm));
      hx = __builtin_fmaxf(hx, __shfl_xor(hx, mm));
      hy = __builtin_fmaxf(hy, __shfl_xor(hy, mm));
      hz = __builtin_fmaxf(hz, __shfl_xor(hz, mm));
    }
    if (lane == i) {  // an empty cluster keeps lo = 3e38: infinitely far from every pick
      blx = lx, bly = ly, blz = lz;
      bhx = hx, bhy = hy, bhz = hz;
    }
  }
  // Slot summaries, kept by lane i for slot i and refreshed whenever the slot is updated:
  //   sval / slow / sx / sown  -- the slot's smallest key (density bits, low), its x and its lane
  //   reach2                   -- squared distance up to which a pick can still CHANGE the slot.
  // (1) exp(-d/t) is exactly 0 beyond d = 104 t.  (2) An increment below half an ulp of the
  // density it is added to leaves the density unchanged: with vmin the smallest density of the
  // slot, every v >= vmin has ulp(v) / 2 >= vmin 2^-25, and an increment is at most
  // 2 exp(-gap^2 / t) (1 + 2^-21) for a pick at squared distance >= gap^2 from the slot's box; so
  // beyond gap^2 = t (ln 2^26 - ln vmin) the update is a no-op for the whole slot.  Late in the
  // run (every region already holds picks) this is a ~3x smaller ball than (1).
  const float far2 = cut2 * 1.001f;
  unsigned sval = 0xffffffffu, slow = 0xffffffffu;
  float sx = 0.f, reach2 = -1.f;
  int sown = 0;
  auto summarize = [&](int i, unsigned tv, unsigned lw, float x) {  // i: static slot index
    const unsigned sm = wave_min_u32(tv);
    unsigned long long eq = __ballot(tv == sm);
    if (__popcll(eq) > 1) {  // equal densities inside the slot: the smaller low wins
      const unsigned cand = tv == sm ? lw : 0xffffffffu;
      eq = __ballot(cand == wave_min_u32(cand));
    }
    const int own = (int)__builtin_ctzll(eq);
    const unsigned o_low = (unsigned)__builtin_amdgcn_readlane((int)lw, own);
    const float o_x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), own));
    const float vmin = __uint_as_float(sm);
    // t ((ln 2^26 - ln vmin) (1 + 1e-3) + 0.01): the margins cover v_log_f32, the rounding of the
    // box test and of d, and sn_expf's <= 2 ulp
    float r2 = far2;
    if (vmin > 1e-30f) {
      const float lg = 18.0219f - 0.693147182f * __builtin_amdgcn_logf(vmin);
      r2 = __builtin_fminf(far2, __builtin_fmaxf(t * (lg * 1.001f + 0.01f), 0.f));
    }
    if (lane == i) {
      sval = sm;
      slow = o_low;
      sx = o_x;
      sown = own;
      reach2 = (blx <= bhx) ? r2 : -1.f;  // empty slot: never near
    }
  };
#pragma unroll
  for (int i = 0; i < PPT; ++i) summarize(i, __float_as_uint(tmp[i]), low[i], px[i]);

  int last = 0;
  if (tid == 0) out[0] = 0;
  // The pick's coordinates travel with the arg-min through LDS (a global read of xyz[last] at
  // the top of each of the 16383 dependent rounds would sit on the critical path).
  __shared__ unsigned wave_val[2][16];
  __shared__ float4 wave_pick[2][16];  // x, y, z, low bits
  const unsigned kBig = __float_as_uint(1e9f);
  const float x0 = p[0], y0 = p[1], z0 = p[2];
  float x1 = x0, y1 = y0, z1 = z0;
  unsigned last_low = 0;  // low bits of point 0

  // Two round loops.  When the cut radius covers most of the cloud (dense regime: t large against
  // the cloud's extent) nearly every slot is updated in every round; summaries per updated slot
  // then cost more than one scan of all slots, and the absorption bound excludes little.
  if (cut2 > 0.5f * diag2) {
    for (int j = 1; j < m; ++j) {
      // which slots of this wave can receive a non-zero update?
      const float gx = __builtin_fmaxf(__builtin_fmaxf(blx - x1, x1 - bhx), 0.f);
      const float gy = __builtin_fmaxf(__builtin_fmaxf(bly - y1, y1 - bhy), 0.f);
      const float gz = __builtin_fmaxf(__builtin_fmaxf(blz - z1, z1 - bhz), 0.f);
      const unsigned mask = (unsigned)__ballot((gx * gx + gy * gy) + gz * gz < far2);
      unsigned mn = 0xffffffffu;  // densities are >= 0: their bit patterns order like the floats
  #pragma unroll
      for (int i = 0; i < PPT; ++i) {
        if ((mask >> i) & 1u) {  // wave-uniform
          const float v = (low[i] == last_low) ? 1e9f : tmp[i];
          const float2 q = reinterpret_cast<const float2 *>(yz)[i * 1024 + tid];
          const float dx = px[i] - x1, dy = q.x - y1, dz = q.y - z1;
          const float d = (dx * dx + dy * dy) + dz * dz;
          const float e = sn_expf_nonpositive(neg_div(d, t, rt, fast_div));
          // points k >= 8192 receive e + e (reference MDS.cu:86-91); doubling is exact
          tmp[i] = v + __builtin_ldexpf(e, (int)(low[i] & 1u));
        }
        mn = umin32(mn, __float_as_uint(tmp[i]));
      }
      // Arg-min of (density, bitrev, k) inside the wave.  Common case: the minimum density is
      // held by exactly one entry, found with one compare per slot; exact ties take the full
      // key comparison.
      const unsigned wm = wave_min_u32(mn);
      int hits = 0, istar = 0;  // scalar: number of entries equal to wm, and their slot
  #pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int c = __popcll(__ballot(__float_as_uint(tmp[i]) == wm));
        hits += c;
        istar += i * c;
      }
      unsigned wl = 0xffffffffu;
      float wx = 0.f;
      int wi = 0;
      bool winner;
      if (hits == 1) {
        winner = mn == wm;
        wi = istar;
  #pragma unroll
        for (int i = 0; i < PPT; ++i)
          if (i == istar) {  // wave-uniform pick of a statically indexed register
            asm volatile("");
            wl = low[i];
            wx = px[i];
          }
      } else {
  #pragma unroll
        for (int i = 0; i < PPT; ++i) {
          const bool lt = __float_as_uint(tmp[i]) == wm && low[i] < wl;
          wl = lt ? low[i] : wl;
          wx = lt ? px[i] : wx;
          wi = lt ? i : wi;
        }
        const unsigned wlmin = wave_min_u32(wl);
        winner = wl == wlmin && wl != 0xffffffffu;  // lows of real points are unique
      }
      const int buf = j & 1;
      if (lane == 0) wave_val[buf][wave] = wm;
      if (winner) {
        const float2 q = reinterpret_cast<const float2 *>(yz)[wi * 1024 + tid];
        wave_pick[buf][wave] = make_float4(wx, q.x, q.y, __uint_as_float(wl));
      }
      __syncthreads();
      // every wave reduces the 16 hand-offs in its own registers
      const int l16 = lane & 15;
      const unsigned v16 = wave_val[buf][l16];
      const float4 pk = wave_pick[buf][l16];
      const unsigned minv = row_min_u32(v16);
      const unsigned lw = v16 == minv ? __float_as_uint(pk.w) : 0xffffffffu;
      const unsigned minl = row_min_u32(lw);
      const int wsel = (int)__builtin_ctzll(__ballot(lw == minl));
      if (__builtin_amdgcn_readfirstlane((int)minv) >= (int)kBig) {
        last = 0;  // nothing below 1e9: the reference's threads all report (1e9, index 0)
        last_low = 0;
        x1 = x0;
        y1 = y0;
        z1 = z0;
      } else {
        last_low = (unsigned)__builtin_amdgcn_readlane((int)lw, wsel);
        last = (int)((last_low >> 1) & 0x7fffu);
        x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pk.x), wsel));
        y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pk.y), wsel));
        z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pk.z), wsel));
      }
      if (tid == 0) out[j] = last;
    }
    return;
  }
  for (int j = 1; j < m; ++j) {
    // which slots of this wave can the pick still change?
    const float gx = __builtin_fmaxf(__builtin_fmaxf(blx - x1, x1 - bhx), 0.f);
    const float gy = __builtin_fmaxf(__builtin_fmaxf(bly - y1, y1 - bhy), 0.f);
    const float gz = __builtin_fmaxf(__builtin_fmaxf(blz - z1, z1 - bhz), 0.f);
    const unsigned mask = (unsigned)__ballot((gx * gx + gy * gy) + gz * gz < reach2);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if ((mask >> i) & 1u) {  // wave-uniform
        const float v = (low[i] == last_low) ? 1e9f : tmp[i];
        const float2 q = reinterpret_cast<const float2 *>(yz)[i * 1024 + tid];
        const float dx = px[i] - x1, dy = q.x - y1, dz = q.y - z1;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float e = sn_expf_nonpositive(neg_div(d, t, rt, fast_div));
        // points k >= 8192 receive e + e (reference MDS.cu:86-91); doubling is exact
        const float nv = v + __builtin_ldexpf(e, (int)(low[i] & 1u));
        tmp[i] = nv;
        summarize(i, __float_as_uint(tmp[i]), low[i], px[i]);
      }
    }
    // Arg-min of (density, bitrev, k) inside the wave = the smallest slot summary.
    const unsigned wm = wave_min_u32(sval);  // lanes >= PPT hold ~0
    unsigned long long eq = __ballot(sval == wm);
    if (__popcll(eq) > 1) {  // equal densities in several slots: the smaller low wins
      const unsigned cand = sval == wm ? slow : 0xffffffffu;
      eq = __ballot(cand == wave_min_u32(cand));
    }
    const int istar = (int)__builtin_ctzll(eq);
    const unsigned wl = (unsigned)__builtin_amdgcn_readlane((int)slow, istar);
    const int wo = __builtin_amdgcn_readlane(sown, istar);
    const int buf = j & 1;
    if (lane == 0) {
      const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), istar));
      const float2 q = reinterpret_cast<const float2 *>(yz)[istar * 1024 + wave * 64 + wo];
      wave_val[buf][wave] = wm;
      wave_pick[buf][wave] = make_float4(wx, q.x, q.y, __uint_as_float(wl));
    }
    __syncthreads();
    // every wave reduces the 16 hand-offs in its own registers
    const int l16 = lane & 15;
    const unsigned v16 = wave_val[buf][l16];
    const float4 pk = wave_pick[buf][l16];
    const unsigned minv = row_min_u32(v16);
    const unsigned lw = v16 == minv ? __float_as_uint(pk.w) : 0xffffffffu;
    const unsigned minl = row_min_u32(lw);
    const int wsel = (int)__builtin_ctzll(__ballot(lw == minl));
    if (__builtin_amdgcn_readfirstlane((int)minv) >= (int)kBig) {
      last = 0;  // nothing below 1e9: the reference's threads all report (1e9, index 0)
      last_low = 0;
      x1 = x0;
      y1 = y0;
      z1 = z0;
    } else {
      last_low = (unsigned)__builtin_amdgcn_readlane((int)lw, wsel);
      last = (int)((last_low >> 1) & 0x7fffu);
      x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pk.x), wsel));
      y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pk.y), wsel));
      z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pk.z), wsel));
    }
    if (tid == 0) out[j] = last;
  }
}

// ---------------------------------------------------------------------------------------
// A cloud on a TEAM of workgroups (mds_dense_team_kernel; the name is round 3's, when only the dense regime took it).
// With one workgroup per cloud a round is bound by the vector ALUs of the ONE compute unit that owns the cloud (4.3 us
// x 16383 rounds = 68 ms at n = 19384 when the cut ball covers the cloud, whatever the batch: 32 of 256 CUs busy at
// B = 32, 4 at the 8-GPU share).  Here G <= 32 workgroups share a cloud: member g keeps PG x nw consecutive groups of
// 64 points of the cluster-sorted cloud (dealt out evenly), updates them, finds its own arg-min exactly as the
// single-workgroup kernel does, and the G candidates meet in global memory:
//   * four 64-bit words per member and EXCHANGE in the member's own cache line, each with the exchange's 6-bit stamp
//     (a word is either entirely of this exchange or not: no fence, no flag + data pair), in a ring of two exchanges:
//     the member's lowest key (density bits : 32 | tie key : 26), its SECOND lowest density, the candidate's
//     coordinates; written by lanes 0-3 of wave 0, polled with agent-scope loads by lanes 0 .. G-1 of wave 0;
//   * every member then replays the same picks among the G candidates (round 6: several exact picks per exchange,
//     see the loop) and applies them in pick order.
//   A member can only run two exchanges ahead of the slowest one (exchange e + 1 needs everybody's words of e), so
//   the two-deep ring never overwrites a word somebody still needs.
// Teams are formed from XCD-local tickets like the auction's (emd.hip): the G workgroups of a cloud sit on one
// XCD whenever the dispatcher allows it, so the words travel through that XCD's L2; any placement is correct.
// Every poll is bounded: on a time-out the launch raises its abort word and the device's sticky word (the next
// sn_mds / sn_emd_* call fails with SN_ETIMEDOUT), and the cloud's whole index row is written as -1 (which
// sn_gather_forward turns into NaN features: never a plausible-looking sample).
// Which clouds take this path: cut^2 > team_ratio x (bounding box diagonal)^2, the predicate by which the
// single-workgroup kernel skips exactly those clouds; since round 6 team_ratio is ~0 for clouds of >= 8192 points on
// teams of >= 8 (every regime; see sn_mds).  Index sequences are identical by construction.
// ---------------------------------------------------------------------------------------
struct MdsTeamCtl {          // zeroed before the launch
  unsigned xticket[8];
  unsigned ticket;
  unsigned abort;
  unsigned pad[22];
  unsigned long long words[1];  // [teams][3][G] x kWordStride: two ring slots + the formation slot, one line each
};
constexpr int kWordStride = 16;  // 128 bytes: every member's word in its own cache line

template <int PG>
__global__ __launch_bounds__(1024) void mds_dense_team_kernel(
    int B, int n, int m, const float *__restrict__ xyz, const int *__restrict__ perm_all,
    const float *__restrict__ bbox, const float *__restrict__ mean_mst_length, int *__restrict__ idxs,
    MdsTeamCtl *ctl, unsigned *sticky, int G, int teams, int xcd_local, unsigned spin_limit, float team_ratio, int nw) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float yz[];  // [PG * nthreads][2], lane private
  __shared__ int s_ticket;
  __shared__ unsigned wave_val[2][16];
  __shared__ unsigned wave_sec[2][16];
  __shared__ float4 wave_pick[2][16];  // x, y, z, low bits
  __shared__ int s_state[2], s_stray;  // 1: a pick, 0: nothing below 1e9 anywhere, -1: a member never answered
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nthreads = nw * 64;  // nw <= 16 waves per member (round 6: as many as its share of the cloud needs)
  if (tid < 32) {  // hand-off entries of waves that do not exist stay neutral: never the minimum
    wave_val[tid >> 4][tid & 15] = 0xffffffffu;
    wave_sec[tid >> 4][tid & 15] = 0xffffffffu;
  }
  if (tid == 0) {
    int t = -1;
    s_stray = 0;
    if (xcd_local & 1) {
      const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);  // HW_REG_XCC_ID
      const int cap = (teams / 8) * G;
      for (int i = 0; i < 9 && t < 0; ++i) {
        const int x = (xcc + i) & 7;
        const int k = (int)__hip_atomic_fetch_add(&ctl->xticket[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k < cap) {
          t = ((k / G) * 8 + x) * G + k % G;
          s_stray = i != 0;  // a slot of another XCD's team
        }
      }
    } else {
      s_stray = 1;
      t = (int)__hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_ticket = t;
  }
  __syncthreads();
  const int ticket = s_ticket;
  if (ticket < 0) return;
  const int b = ticket / G, g = ticket % G;
  if (b >= B || b >= teams) return;
  if ((xcd_local & 2) && ticket == 1) return;  // test knob (SN_MDS_DIAG=8): a team member that never arrives
  const float *__restrict__ p = xyz + (size_t)b * n * 3;
  const int *__restrict__ perm = perm_all + (size_t)b * n;
  int *__restrict__ out = idxs + (size_t)b * m;
  const float mml = mean_mst_length[b];
  const float t = (float)(5.0 * (double)mml * (double)mml);
  const float cut2 = 104.0f * t * 1.0001f;
  const float rt = 1.0f / t;
  float diag2 = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ext = bbox[b * 6 + 3 + a] - bbox[b * 6 + a];
    diag2 += ext * ext;
  }
  if (!(cut2 > team_ratio * diag2)) return;  // not dense enough: the single-workgroup kernel owns this cloud
  const bool fast_div = t >= 0x1p-40f && t <= 0x1p40f && diag2 * rt < 0x1p100f;
  const float far2 = cut2 * 1.001f;

  float px[PG], tmp[PG];
  unsigned low[PG];
  float blx = 3e38f, bly = 3e38f, blz = 3e38f, bhx = -3e38f, bhy = -3e38f, bhz = -3e38f;
#pragma unroll
  for (int i = 0; i < PG; ++i) {
    const int s = (((g * PG + i) * nw + wave) << 6) + lane;  // sorted position: the member owns PG x nw consecutive groups of 64
    const bool valid = s < n;
    const int k = valid ? perm[s] : 0;
    const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
    px[i] = valid ? x : 0.f;
    yz[2 * (i * nthreads + tid) + 0] = valid ? y : 0.f;
    yz[2 * (i * nthreads + tid) + 1] = valid ? z : 0.f;
    tmp[i] = valid ? 0.f : 1e9f;
    low[i] = valid ? ((__brev((unsigned)(k & 1023)) >> 22) << 16) | ((unsigned)k << 1) | (k >= 8192 ? 1u : 0u)
                   : 0xffffffffu;
    float lx = valid ? x : 3e38f, ly = valid ? y : 3e38f, lz = valid ? z : 3e38f;
    float hx = valid ? x : -3e38f, hy = valid ? y : -3e38f, hz = valid ? z : -3e38f;
    for (int mm = 1; mm < 64; mm <<= 1) {
      lx = __builtin_fminf(lx, __shfl_xor(lx, mm));
      ly = __builtin_fminf(ly, __shfl_xor(ly, mm));
      lz = __builtin_fminf(lz, __shfl_xor(lz, mm));
      hx = __builtin_fmaxf(hx, __shfl_xor(hx, mm));
      hy = __builtin_fmaxf(hy, __shfl_xor(hy, mm));
      hz = __builtin_fmaxf(hz, __shfl_xor(hz, mm));
    }
    if (lane == i) {
      blx = lx, bly = ly, blz = lz;
      bhx = hx, bhy = hy, bhz = hz;
    }
  }
  int last = 0;
  if (tid == 0 && g == 0) out[0] = 0;
  const unsigned kBig = __float_as_uint(1e9f);
  const float x0 = p[0], y0 = p[1], z0 = p[2];
  float x1 = x0, y1 = y0, z1 = z0;
  unsigned last_low = 0;  // low bits of point 0
  unsigned long long *words = ctl->words + (size_t)b * 3 * G * kWordStride;   // [2 ring slots + 1 formation slot][G]
  // Is the whole team on one XCD?  Every member says whether it took a slot of another XCD's team (agent-scope
  // words, round stamp 63 -- the rounds start at 1), then everybody knows.
  bool loc = false;
  {
    if (wave == 0) {
      unsigned long long *slot = words + ((size_t)2 * G) * kWordStride;
      if (lane == 0)
        __hip_atomic_store(&slot[(size_t)g * kWordStride], 63ull | ((unsigned long long)s_stray << 6), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long w = 63ull | ((unsigned long long)s_stray << 6);
      if (lane < G && lane != g) {
        unsigned spins = 0;
        for (;;) {
          w = __hip_atomic_load(&slot[(size_t)lane * kWordStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((w & 63ull) == 63ull) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > spin_limit) {
            __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sticky) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          if ((spins & 1023u) == 0u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
      }
      const bool missing = lane < G && (w & 63ull) != 63ull;
      const bool strayed = lane < G && ((w >> 6) & 1ull) != 0ull;
      // wave-level votes BEFORE the one-lane branch: inside `if (lane == 0)` a vote only sees lane 0
      const bool any_missing = __any(missing), any_strayed = __any(strayed);
      if (lane == 0) s_state[0] = any_missing ? -1 : (any_strayed ? 0 : 1);
    }
    __syncthreads();
    if (s_state[0] < 0) {  // the team never formed: nothing of this cloud is a result (-1: sn_gather_* -> NaN)
      if (g == 0)
        for (int e = tid; e < m; e += nthreads) out[e] = -1;
      return;
    }
    loc = s_state[0] == 1;
    __syncthreads();
  }

  // ---- rounds: SEVERAL PICKS PER EXCHANGE (round 6) ----------------------------------------------------------------
  // A round of rounds 3-5 was: update -> arg-min in the workgroup -> exchange of one word per member -> the pick's
  // coordinates: 1.7 us, of which 0.84 us exchange + coordinates (profiles/r06_a_mds_dense_stamps.txt).  Now an exchange
  // carries, per member, its lowest candidate WITH its coordinates and its SECOND lowest density, and every member
  // replays the same deterministic little auction among the G candidates:
  //   pick 1 = the smallest key (density, tie key) of the G candidates -- the team's arg-min, as before;
  //   after a pick, a candidate's new density needs only its old density and the pick's coordinates: the replay adds
  //     exactly what the owner will add (same operands, same operations: bit-identical), the picked one goes to 1e9;
  //   pick q + 1 = the smallest replayed key -- PROVIDED it is strictly below `bound`, the smallest of the members'
  //     second-lowest densities at the time of the exchange: densities only grow, so every point that is not a
  //     candidate still has a density >= its member's second-lowest >= bound, and the replayed minimum is the arg-min
  //     of the whole cloud (an equal density outside the set could win on the tie key: strict '<' stops there).
  // The accepted picks are then applied in pick order by every member (each density takes its increments in the
  // reference's order: the per-round rounding of (float)((double)temp + w exp) is replayed exactly), and the next
  // arg-min runs once per exchange.  On the dense regime's data the test accepts 2.5-4.6 picks per exchange
  // (tools/sim/mds_dense_staleness.py: the lowest densities belong to points far from every pick so far, which a new pick
  // far away hardly changes).  The index sequence is the reference's by construction; tests/test_mds.py and
  // tests/test_fullsize.py compare it with the oracle's.
  constexpr int kMaxQ = 16;  // picks per exchange at most
  __shared__ float4 s_picks[kMaxQ];  // x, y, z, low bits
  __shared__ int s_npick;
  if (tid == 0) s_picks[0] = make_float4(x0, y0, z0, __uint_as_float(0u));  // the first sample: point 0 (idx[0] = 0)
  __syncthreads();
  (void)last;
#ifdef SN_MDS_STAMPS
  const bool stamping = b == 0 && g == 0 && wave == 0;
  long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  long long st_tk = (long long)__builtin_amdgcn_s_memrealtime();
  unsigned st_ex = 0;
#endif
  int j = 1, npick = 1;
  for (unsigned ex = 1; j < m; ++ex) {
    // the picks decided by the previous exchange (first: point 0), in pick order
    for (int q = 0; q < npick; ++q) {  // uniform
      const float4 pq = s_picks[q];
      x1 = pq.x;
      y1 = pq.y;
      z1 = pq.z;
      last_low = __float_as_uint(pq.w);
      const float gx = __builtin_fmaxf(__builtin_fmaxf(blx - x1, x1 - bhx), 0.f);
      const float gy = __builtin_fmaxf(__builtin_fmaxf(bly - y1, y1 - bhy), 0.f);
      const float gz = __builtin_fmaxf(__builtin_fmaxf(blz - z1, z1 - bhz), 0.f);
      const unsigned mask = (unsigned)__ballot((gx * gx + gy * gy) + gz * gz < far2);
#pragma unroll
      for (int i = 0; i < PG; ++i) {
        if ((mask >> i) & 1u) {  // wave-uniform
          const float v = (low[i] == last_low) ? 1e9f : tmp[i];
          const float2 qq = reinterpret_cast<const float2 *>(yz)[i * nthreads + tid];
          const float dx = px[i] - x1, dy = qq.x - y1, dz = qq.y - z1;
          const float d = (dx * dx + dy * dy) + dz * dz;
          const float e = sn_expf_nonpositive(neg_div(d, t, rt, fast_div));
          tmp[i] = v + __builtin_ldexpf(e, (int)(low[i] & 1u));
        }  // (the pick's own slot is always inside the ball: its 1e9 mark is never skipped)
      }
    }
    MDS_STAMP(0)
    // this lane's smallest and second smallest density (equal densities count), then the wave's
    unsigned mn = 0xffffffffu, m2 = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const unsigned v = __float_as_uint(tmp[i]);
      m2 = umin32(m2, v > mn ? v : mn);
      mn = umin32(mn, v);
    }
    // arg-min of (density, low) inside the wave: the full key comparison (PG is small)
    const unsigned wm = wave_min_u32(mn);
    unsigned wl = 0xffffffffu;
    float wx = 0.f;
    int wi = 0;
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const bool lt = __float_as_uint(tmp[i]) == wm && low[i] < wl;
      wl = lt ? low[i] : wl;
      wx = lt ? px[i] : wx;
      wi = lt ? i : wi;
    }
    const unsigned wlmin = wave_min_u32(wl);
    const bool winner = wl == wlmin && wl != 0xffffffffu;  // lows of real points are unique
    // the wave's second smallest density: the winner lane contributes its own second, every other lane its smallest
    const unsigned wsec = wave_min_u32(winner ? m2 : mn);
    const int buf = (int)(ex & 1u);
    if (lane == 0) {
      wave_val[buf][wave] = wm;
      wave_sec[buf][wave] = wsec;
    }
    if (winner) {
      const float2 qq = reinterpret_cast<const float2 *>(yz)[wi * nthreads + tid];
      wave_pick[buf][wave] = make_float4(wx, qq.x, qq.y, __uint_as_float(wl));
    }
    __syncthreads();
    MDS_STAMP(1)
    if (wave == 0) {  // ONE wave per workgroup talks to the others (sixteen pollers per workgroup on the same lines
                      // slowed every store down); the rest of the workgroup waits at the barrier below
      const int l16 = lane & 15;
      const unsigned v16 = wave_val[buf][l16], s16 = wave_sec[buf][l16];
      const float4 pk = wave_pick[buf][l16];
      const unsigned minv = row_min_u32(v16);
      const unsigned lw = v16 == minv ? __float_as_uint(pk.w) : 0xffffffffu;
      const unsigned minl = row_min_u32(lw);
      const int wsel = (int)__builtin_ctzll(__ballot(lw == minl));  // < 16: the wave that holds the member's candidate
      const unsigned msec = row_min_u32(l16 == wsel ? s16 : v16);   // the member's second smallest density
      // this workgroup's candidate: (minv, minl) + coordinates; "nothing below 1e9" travels as (>= kBig, all ones)
      const unsigned my_val = (unsigned)__builtin_amdgcn_readfirstlane((int)minv);
      const unsigned my_low = my_val >= kBig ? 0x3ffffffu : ((unsigned)__builtin_amdgcn_readfirstlane((int)minl) & 0x3ffffffu);
      const unsigned my_sec = (unsigned)__builtin_amdgcn_readfirstlane((int)msec);
      const unsigned xb = (unsigned)__builtin_amdgcn_readlane(__float_as_int(pk.x), wsel);
      const unsigned yb = (unsigned)__builtin_amdgcn_readlane(__float_as_int(pk.y), wsel);
      const unsigned zb = (unsigned)__builtin_amdgcn_readlane(__float_as_int(pk.z), wsel);
      const unsigned stamp = ex & 63u;
      // four 64-bit words in the member's line, EACH with the exchange's stamp in its low six bits (a word is either
      // entirely of this exchange or not: no fence, no flag + data pair):
      //   W0 = density : 32 | low key : 26 | stamp     W1 = second density : 32 | x[31:6] : 26 | stamp
      //   W2 = y : 32 | z[31:6] : 26 | stamp            W3 = x[5:0] << 12 | z[5:0] << 6 | stamp
      unsigned long long *slot = words + ((size_t)buf * G) * kWordStride;
      if (lane < 4) {
        const unsigned long long w0 = ((unsigned long long)my_val << 32) | ((unsigned long long)my_low << 6) | stamp;
        const unsigned long long w1 = ((unsigned long long)my_sec << 32) | (unsigned long long)(xb & ~63u) | stamp;
        const unsigned long long w2 = ((unsigned long long)yb << 32) | (unsigned long long)(zb & ~63u) | stamp;
        const unsigned long long w3 = ((unsigned long long)(xb & 63u) << 12) | ((unsigned long long)(zb & 63u) << 6) | stamp;
        const unsigned long long mine = lane == 0 ? w0 : (lane == 1 ? w1 : (lane == 2 ? w2 : w3));
        // a team on ONE XCD keeps its words in that XCD's L2 (plain store; the pollers' coherent loads meet it
        // there); an agent-scope store goes out to the fabric and takes the line with it
        if (loc)
          __hip_atomic_store(&slot[(size_t)g * kWordStride + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
          __hip_atomic_store(&slot[(size_t)g * kWordStride + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // lane l < G: member l's candidate
      unsigned cv = my_val, cl = my_low, csec = my_sec;
      float cx = __uint_as_float(xb), cy = __uint_as_float(yb), cz = __uint_as_float(zb);
      bool stale = false;
      if (lane < G && lane != g) {
        const unsigned long long *src = &slot[(size_t)lane * kWordStride];
        unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        unsigned spins = 0;
        for (;;) {
          w0 = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w3 = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((((unsigned)w0 ^ stamp) | ((unsigned)w1 ^ stamp) | ((unsigned)w2 ^ stamp) | ((unsigned)w3 ^ stamp)) << 26 == 0u) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++spins & 1023u) == 0u) {
            if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { stale = true; break; }
            if (spins > spin_limit) {
              __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (sticky) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              stale = true;
              break;
            }
          }
        }
        cv = (unsigned)(w0 >> 32);
        cl = (unsigned)(w0 >> 6) & 0x3ffffffu;
        csec = (unsigned)(w1 >> 32);
        cx = __uint_as_float(((unsigned)w1 & ~63u) | ((unsigned)(w3 >> 12) & 63u));
        cy = __uint_as_float((unsigned)(w2 >> 32));
        cz = __uint_as_float(((unsigned)w2 & ~63u) | ((unsigned)(w3 >> 6) & 63u));
      }
      MDS_STAMP(2)
      if (lane >= G) {
        cv = 0xffffffffu;
        cl = 0x3ffffffu;
        csec = 0xffffffffu;
      }
      const bool any_stale = __any(stale);  // voted by the whole wave, not inside a one-lane branch
      // what no member published: every such point's density is >= its member's second smallest >= bound
      const unsigned bound = wave_min_u32(csec);
      // (Measured and not kept, round 6: the G x G increments the candidates would receive from each other computed
      // up front by the wave's 64 lanes -- four independent evaluations per lane -- and the replay reduced to a
      // minimum, an LDS row read and an add per pick: 481 ns of replay per pick against 418 with the chain below,
      // profiles/r06_e_mds_dense_stamps_pair_matrix_not_kept.txt.  The same matrix spread over ALL the workgroup's waves
      // between two more workgroup barriers: 15.1 against 13.8 ms per call at 4 clouds, 11.1 against 10.6 in the surface
      // regime, profiles/r06_k_mds_pairs_all_waves_not_kept.txt.  The chain below stays.)
      int np = 0, state = 1;
      const int room = m - j;  // picks still wanted (>= 1)
      for (int q = 0; q < kMaxQ; ++q) {  // uniform
        const unsigned mv = wave_min_u32(cv);
        if (mv >= kBig) {  // nothing below 1e9 anywhere: the reference's threads all report (1e9, index 0)
          if (q == 0) {
            if (lane == 0) s_picks[0] = make_float4(x0, y0, z0, __uint_as_float(0u));
            np = 1;
            state = 0;
          }
          break;
        }
        if (q > 0 && !(mv < bound)) break;  // a point outside the candidate set could be the arg-min
        unsigned long long eq = __ballot(cv == mv);
        if (__popcll(eq) > 1) {  // equal densities: the smaller tie key wins
          const unsigned ml = wave_min_u32(cv == mv ? cl : 0xffffffffu);
          eq = __ballot(cv == mv && cl == ml);
        }
        const int wq = (int)__builtin_ctzll(eq);
        const float qx = lane_f(cx, wq), qy = lane_f(cy, wq), qz = lane_f(cz, wq);
        const unsigned ql = (unsigned)__builtin_amdgcn_readlane((int)cl, wq);
        if (lane == 0) s_picks[q] = make_float4(qx, qy, qz, __uint_as_float(ql));
        np = q + 1;
        if (np >= room) break;
        // the candidates after this pick: exactly the owner's update (same operands, same operations).
        // (Measured and not kept: skipping the exponential's chain when no candidate lies inside the pick's cut ball --
        // the rule on surface clouds: -3 % there, +3 % in the dense regime, profiles/r06_i_mds_*.)
        const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float e = sn_expf_nonpositive(neg_div(d, t, rt, fast_div));
        const float nv = __uint_as_float(cv) + __builtin_ldexpf(e, (int)(cl & 1u));
        cv = lane >= G ? 0xffffffffu : (lane == wq ? kBig : __float_as_uint(nv));
      }
      if (lane == 0) {
        s_npick = np;
        s_state[buf] = any_stale ? -1 : state;
      }
#ifdef SN_MDS_STAMPS
      if (stamping) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
      MDS_STAMP(3)
    }
    __syncthreads();
    MDS_STAMP(4)
#ifdef SN_MDS_STAMPS
    ++st_ex;
#endif
    const int state = s_state[buf];
    if (state < 0) {  // a member never answered (workgroup-uniform): the WHOLE row becomes -1 -- a partial sequence
      if (g == 0)     // would look like a result; sn_gather_forward turns -1 into NaN, the next sn_mds call fails
        for (int e = tid; e < m; e += nthreads) out[e] = -1;
      return;
    }
    npick = s_npick;
    if (tid < npick && g == 0) {  // state 0: nothing below 1e9 anywhere -> (1e9, index 0)
      const unsigned lwq = __float_as_uint(s_picks[tid].w);
      out[j + tid] = state ? (int)((lwq >> 1) & 0x7fffu) : 0;
    }
    j += npick;
  }
#ifdef SN_MDS_STAMPS
  if (stamping && lane == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_mds_stamps[i], (unsigned long long)st_acc[i]);
    atomicAdd(&g_mds_stamps[5], (unsigned long long)(m - 1));
    atomicAdd(&g_mds_stamps[6], (unsigned long long)st_ex);
  }
#endif
}

// generic fallback for clouds that do not fit the register budget: state in global memory
__global__ __launch_bounds__(1024) void mds_kernel_generic(int n, int m,
                                                           const float *__restrict__ xyz,
                                                           const float *__restrict__ mean_mst_length,
                                                           float *__restrict__ temp,
                                                           int *__restrict__ idxs, int lg) {
#pragma clang fp contract(off)
  __shared__ unsigned long long wave_key[2][16];
  const int b = blockIdx.x, bs = blockDim.x, tid = threadIdx.x;
  const float *__restrict__ p = xyz + (size_t)b * n * 3;
  float *__restrict__ tp = temp + (size_t)b * n;
  int *__restrict__ out = idxs + (size_t)b * m;
  const float mml = mean_mst_length[b];
  const float t = (float)(5.0 * (double)mml * (double)mml);
  const unsigned rev = lg ? (__brev((unsigned)tid) >> (32 - lg)) : 0u;
  const int wave_width = bs < 64 ? bs : 64;
  const int nwaves = (bs + 63) / 64;
  for (int k = tid; k < n; k += bs) tp[k] = 0.f;
  int last = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = p[last * 3 + 0], y1 = p[last * 3 + 1], z1 = p[last * 3 + 2];
    float bestv = 1e9f;
    int bk = 0;
    for (int k = tid; k < n; k += bs) {
      float v = (k == last) ? 1e9f : tp[k];
      const float dx = p[k * 3 + 0] - x1, dy = p[k * 3 + 1] - y1, dz = p[k * 3 + 2] - z1;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const float e = sn_expf(-d / t);
      v = v + (k < 8192 ? e : e + e);
      tp[k] = v;
      if (v < bestv) {
        bestv = v;
        bk = k;
      }
    }
    // key: density | bitrev(tid) (10 bits) | k (22 bits)
    unsigned long long key = ((unsigned long long)__float_as_uint(bestv) << 32) |
                             (unsigned long long)((rev << 22) | (unsigned)bk);
    for (int s = 1; s < wave_width; s <<= 1) {
      const unsigned long long o = shfl_xor_u64(key, s, wave_width);
      key = o < key ? o : key;
    }
    if (nwaves > 1) {
      const int buf = j & 1;
      if ((tid & 63) == 0) wave_key[buf][tid >> 6] = key;
      __syncthreads();
      const int l = tid & 15;
      key = l < nwaves ? wave_key[buf][l] : ~0ull;
      for (int s = 1; s < 16; s <<= 1) {
        const unsigned long long o = shfl_xor_u64(key, s, 16);
        key = o < key ? o : key;
      }
    }
    last = (unsigned)(key >> 32) >= __float_as_uint(1e9f) ? 0 : (int)((unsigned)key & 0x3fffffu);
    if (tid == 0) out[j] = last;
  }
}

__global__ __launch_bounds__(256) void gather_fwd_kernel(int c, int n, int m,
                                                         const float *__restrict__ feat,
                                                         const int *__restrict__ idx,
                                                         float *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e % m);
    const long bc = e / m;
    const long b = bc / c;
    const int k = idx[b * m + j];
    // an index outside the cloud is the sampler's "no result" marker (-1 after a time-out): NaN, not a wild read
    out[e] = (unsigned)k < (unsigned)n ? feat[bc * n + k] : __builtin_nanf("");
  }
}

__global__ __launch_bounds__(256) void gather_bwd_kernel(int c, int n, int m,
                                                         const float *__restrict__ grad_out,
                                                         const int *__restrict__ idx,
                                                         float *__restrict__ grad_feat,
                                                         long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e % m);
    const long bc = e / m;
    const long b = bc / c;
    const int k = idx[b * m + j];
    if ((unsigned)k < (unsigned)n) unsafeAtomicAdd(&grad_feat[bc * n + k], grad_out[e]);
  }
}

static int lin_blocks(long total) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

// dynamic y/z slots (8 B x 1024 lanes x slots) + 768 B of static hand-off storage <= 160 KiB
static bool mds_use_clustered(int n) {
  return n >= 2048 && (size_t)((n + 1023) / 1024) * 8192 + 1024 <= 160 * 1024;
}

constexpr size_t kMdsTeamCtlBytes = 128 + 128 * 3 * 1024;  // header + exchange lines of up to 1024 team members

extern "C" size_t sn_mds_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  if (mds_use_clustered(n))  // perm + cell ids + cell offsets + bounding boxes + the dense-regime teams' control block
    return sn::align_up((size_t)b * n * 4, 256) * 2 + (size_t)b * kSortCells * 4 + sn::align_up(256 * (size_t)b, 256) +
           kMdsTeamCtlBytes;
  int bs = 1;
  while (bs * 2 <= n && bs < 1024) bs *= 2;
  return (n + bs - 1) / bs <= 24 ? 0 : (size_t)b * n * 4;
}

extern "C" int sn_mds(const float *xyz, int b, int n, int m, const float *mean_mst_length,
                      int *idx, void *workspace, size_t workspace_bytes, void *stream) {
  SN_REQUIRE(xyz && mean_mst_length && idx, "sn_mds: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1 && m >= 1, "sn_mds: need b,n,m >= 1 (got %d,%d,%d)", b, n, m);
  SN_REQUIRE(m <= n, "sn_mds: npoint (%d) must not exceed the cloud size (%d)", m, n);
  SN_REQUIRE(n < (1 << 22), "sn_mds: cloud too large");
  int bs = 1, lg = 0;
  while (bs * 2 <= n && bs < 1024) {
    bs *= 2;
    ++lg;
  }
  const int ppt = (n + bs - 1) / bs;
  hipStream_t s = sn::as_stream(stream);
  if (sn::prof_enabled()) sn::prof_begin("mds", s);
  if (mds_use_clustered(n)) {
    SN_REQUIRE(workspace && workspace_bytes >= sn_mds_workspace_bytes(b, n),
               "sn_mds: workspace too small (%zu < %zu)", workspace_bytes, sn_mds_workspace_bytes(b, n));
    char *w = static_cast<char *>(workspace);
    int *perm = reinterpret_cast<int *>(w); w += sn::align_up((size_t)b * n * 4, 256);
    int *cell_of = reinterpret_cast<int *>(w); w += sn::align_up((size_t)b * n * 4, 256);
    int *hist = reinterpret_cast<int *>(w); w += (size_t)b * kSortCells * 4;
    float *bbox = reinterpret_cast<float *>(w); w += sn::align_up(256 * (size_t)b, 256);
    MdsTeamCtl *tctl = reinterpret_cast<MdsTeamCtl *>(w);
    SN_REQUIRE(cloud_sort(b, n, xyz, bbox, hist, cell_of, perm, s) == 0, "sn_mds: cannot size the sort kernel's LDS");
    // Dense-regime clouds go to a team of G workgroups each (mds_dense_team_kernel); the one-workgroup kernel below
    // then skips them.  G = the largest power of two <= 16 such that every cloud's team fits one XCD's share of the
    // compute units and the whole grid is resident (the members wait for each other); SN_MDS_G overrides (1: off).
    // Measured at n = 19384 -> 16384 (profiles/r03_*_mds_teams.txt): a team round costs ~1.9 us (G = 8) / 1.7 us
    // (G = 16) whatever the regime, the one-workgroup kernel 1.1 us (surface) ... 4.3 us (cut ball = the cloud);
    // the two cross where the cut ball's squared radius is ~0.15 of the box diagonal's.
    int team_g = 1, team_slots = 0;
    float team_ratio_eff = 0.f;   // = team_ratio, or ~0 where teams serve every regime (set below)
    // a cloud goes to a team when its cut ball's squared radius exceeds this fraction of the squared diagonal of
    // its bounding box (SN_MDS_RATIO; measured cross-over)
    // (0.12 in rounds 3-4, from uniform cubes at chosen mean MST lengths.  The first sampler call of an UNTRAINED generator
    // -- the decoder's cube + the partial input, ratio between 0.09 and 0.12 -- ran 44 ms in the one-workgroup kernel where a
    // team takes 27: config 4 at random init 110.7 -> 93.9 ms per step, config 5 179 -> 134 with 0.075; 0.09 still leaves some
    // of config 5's clouds behind (171); a surface cloud at mean MST length 0.02 sits between 0.06 and 0.075 and is better
    // off on one workgroup (23.5 ms against 26-28 on a team), a trained generator's clouds are far below either.
    // profiles/r05_g_mds_team_ratio.txt)
    static const float team_ratio = [] { const char *e = getenv("SN_MDS_RATIO"); const float v = e ? (float)atof(e) : 0.075f; return v > 0.f ? v : 0.075f; }();
    team_ratio_eff = team_ratio;
    {
      int dev = 0, cus = 0;
      SN_HIP(hipGetDevice(&dev));
      if (const int rc = sn::check_sticky(dev, "sn_mds")) return rc;
      SN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      static const int gmax = [] { const char *e = getenv("SN_MDS_G"); const int v = e ? atoi(e) : 32; return v >= 1 && v <= 32 ? v : 32; }();
      // A member owns PG x nw consecutive groups of 64 sorted points (nw <= 16 waves, PG register slots per lane): the
      // cloud's ceil(n / 64) groups are dealt out evenly.  (Rounds 3-5 dealt out whole slots of 1024 points: at n = 19384
      // and G = 16 ten members held two slots each and six held nothing.)  G = the largest power of two <= 32 such that
      // every cloud's team fits one XCD's share of the compute units and a member still has four waves of points.
      const int groups = (n + 63) / 64;
      if (cus >= 64 && cus % 8 == 0 && ppt >= 2) {
        const int per = cus / 8, tpx = (b + 7) / 8;
        int g = 1;
        while (g * 2 * tpx <= per && g * 2 <= gmax && g * 2 * 4 <= groups) g *= 2;
        team_g = g;
        team_slots = 8 * tpx;
      }
      if (team_g >= 2 && sn::capturing(s)) team_g = 1;  // under graph capture: the one-workgroup kernel only
      // Round 6: with several picks per exchange a team of >= 8 members beats the one-workgroup kernel in EVERY regime of
      // a SpareNet-sized cloud -- surface regime (mml 0.0085, 19384 -> 16384) 17.7 -> 10.6 ms at <= 8 clouds, 17.9 ->
      // 15.4 at 32; surface-like clouds 18.0-23.5 -> 10.6-12.2 (profiles/r06_h_mds_team_everywhere.txt) -- so such
      // clouds all go to teams; smaller clouds and teams keep the measured cross-over above.
      if (team_g >= 8 && n >= 8192 && !getenv("SN_MDS_RATIO")) team_ratio_eff = 1e-30f;
      if (team_g >= 2) {
        const int per_member = (groups + team_g - 1) / team_g;        // groups of 64 points a member owns
        const int pg = (per_member + 15) / 16;                        // register slots per lane
        const int nw = (per_member + pg - 1) / pg;                    // waves per member (<= 16)
        unsigned *sticky = sn::sticky_device_word(dev);
        // SN_MDS_DIAG=8 (tests): the second member of cloud 0's team leaves at once and the polls give up early
        const char *dg = SN_KNOB("SN_MDS_DIAG");
        const bool park = dg && atoi(dg) == 8;
        SN_REQUIRE(team_slots * team_g <= 1024 && pg <= 10, "sn_mds: unexpected team geometry");
        SN_HIP(hipMemsetAsync(tctl, 0, 128 + 128 * 3 * (size_t)team_slots * team_g, s));
        sn::PersistentLaunch chain(dev, s);  // never beside another team-waiting launch of this process (common.hpp)
#define SN_MDST(P)                                                                                          \
  {                                                                                                         \
    SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_dense_team_kernel<P>),                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));             \
    mds_dense_team_kernel<P><<<team_slots * team_g, nw * 64, (size_t)P * nw * 64 * 8, s>>>(                 \
        b, n, m, xyz, perm, bbox, mean_mst_length, idx, tctl, sticky, team_g, team_slots, park ? 3 : 1,     \
        park ? 1u << 14 : 1u << 24, team_ratio_eff, nw);                                                    \
  }
        if (pg <= 1) SN_MDST(1)
        else if (pg <= 2) SN_MDST(2)
        else if (pg <= 3) SN_MDST(3)
        else if (pg <= 4) SN_MDST(4)
        else if (pg <= 5) SN_MDST(5)
        else if (pg <= 6) SN_MDST(6)
        else if (pg <= 8) SN_MDST(8)
        else SN_MDST(10)
#undef SN_MDST
      }
    }
    const float skip_ratio = team_g >= 2 ? team_ratio_eff : 0.f;
    const size_t lds = (size_t)ppt * 1024 * 8;
#define SN_MDSC(P)                                                                               \
  {                                                                                              \
    /* every call: the attribute belongs to the CURRENT device (several devices per process under    \
       DataParallel), it is not a per-process fact */                                            \
    SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_clustered_kernel<P>),         \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));  \
    mds_clustered_kernel<P><<<b, 1024, lds, s>>>(n, m, xyz, perm, bbox, mean_mst_length, idx, skip_ratio); \
  }
    // exact slot counts near the register limit (19 at SpareNet's n = 19384): every unused
    // slot costs three VGPRs and the 1024-lane workgroup only has 128 per lane
    if (ppt <= 2) SN_MDSC(2)
    else if (ppt <= 4) SN_MDSC(4)
    else if (ppt <= 8) SN_MDSC(8)
    else if (ppt <= 12) SN_MDSC(12)
    else if (ppt <= 16) SN_MDSC(16)
    else if (ppt == 17) SN_MDSC(17)
    else if (ppt == 18) SN_MDSC(18)
    else SN_MDSC(19)
#undef SN_MDSC
    if (sn::prof_enabled()) sn::prof_end("mds", s);
    return sn::launch_status("sn_mds");
  }
#define SN_MDS(P) mds_kernel<P, 0, 1024><<<b, 1024, 0, s>>>(n, m, xyz, mean_mst_length, idx, lg)
#define SN_MDS_Z(P, C) \
  mds_kernel<P, C, 1024><<<b, 1024, (size_t)n * 4 * C, s>>>(n, m, xyz, mean_mst_length, idx, lg)
  if (bs < 1024) mds_kernel<2, 0, 0><<<b, bs, 0, s>>>(n, m, xyz, mean_mst_length, idx, lg);
  else if (ppt <= 2) SN_MDS(2);
  else if (ppt <= 4) SN_MDS(4);
  else if (ppt <= 8) SN_MDS(8);
  else if (ppt <= 12) SN_MDS(12);
  else if (ppt <= 16) SN_MDS(16);
  else if (ppt <= 20 && (size_t)n * 8 + 1024 <= 160 * 1024) {
    // y,z of 20480 points = 160 KiB minus the hand-off slots: opt in to the large LDS carve
    SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_kernel<20, 2, 1024>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    SN_MDS_Z(20, 2);
  } else if (ppt <= 24) {
    SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_kernel<24, 1, 1024>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    SN_MDS_Z(24, 1);
  }
  else {
    SN_REQUIRE(workspace && workspace_bytes >= sn_mds_workspace_bytes(b, n),
               "sn_mds: workspace too small for n=%d", n);
    mds_kernel_generic<<<b, bs, 0, s>>>(n, m, xyz, mean_mst_length,
                                        static_cast<float *>(workspace), idx, lg);
  }
#undef SN_MDS
#undef SN_MDS_Z
  if (sn::prof_enabled()) sn::prof_end("mds", s);
  return sn::launch_status("sn_mds");
}

extern "C" int sn_gather_forward(const float *feat, const int *idx, int b, int c, int n, int m,
                                 float *out, void *stream) {
  SN_REQUIRE(feat && idx && out, "sn_gather_forward: null pointer");
  SN_REQUIRE(b >= 1 && c >= 1 && n >= 1 && m >= 1, "sn_gather_forward: bad sizes");
  const long total = (long)b * c * m;
  gather_fwd_kernel<<<lin_blocks(total), 256, 0, sn::as_stream(stream)>>>(c, n, m, feat, idx, out,
                                                                          total);
  return sn::launch_status("sn_gather_forward");
}

extern "C" int sn_gather_backward(const float *grad_out, const int *idx, int b, int c, int n,
                                  int m, float *grad_feat, void *stream) {
  SN_REQUIRE(grad_out && idx && grad_feat, "sn_gather_backward: null pointer");
  SN_REQUIRE(b >= 1 && c >= 1 && n >= 1 && m >= 1, "sn_gather_backward: bad sizes");
  hipStream_t s = sn::as_stream(stream);
  SN_HIP(hipMemsetAsync(grad_feat, 0, (size_t)b * c * n * 4, s));
  const long total = (long)b * c * m;
  gather_bwd_kernel<<<lin_blocks(total), 256, 0, s>>>(c, n, m, grad_out, idx, grad_feat, total);
  return sn::launch_status("sn_gather_backward");
}
