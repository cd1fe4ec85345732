// mds.hip -- minimum density sampling + gather for MI355X (gfx950).
//
// Reference: cuda/MDS/MDS_cuda.cu:91-211 (sampling), :29-79 (gather fwd/bwd),
// binding cuda/MDS/MDS.cpp:54-135.  Semantics: oracle/mds.c.  Greedy: every round
// adds exp(-d/t) (x2 for k >= 8192) of the last pick to every point's density and
// picks the arg-min; ties resolve to argmin (bitrev(k mod bs), k), the order the
// reference's reduction tree induces.  The exponential is sn_expf (shared with the
// oracle, include/sn_expf.h).
//
// MI355X design: the op is 16383 DEPENDENT rounds per cloud, so the lever is the
// latency of one round.  One workgroup (bs <= 1024 lanes = 16 waves = one CU) owns a
// cloud and keeps the WHOLE state in registers: lane tid owns points tid, tid+bs, ...
// (coordinates + density, 4 VGPRs per point, 19 points per lane at n = 19384) -- the
// reference re-reads xyz and read-modify-writes `temp` in global memory every round.
// Arg-min = per-lane scan, wave64 xor-butterfly on a packed 64-bit key
// (density bits << 32 | bitrev(tid) << 8 | slot), one LDS hand-off between the 16
// waves with double buffering => ONE barrier per round (reference: 11).
// (float)((double)temp + w) equals the plain fp32 sum for every pair of floats
// (the double sum is exact unless w < ulp(temp)/32, where both round to temp), so
// the accumulation stays in fp32.
#include "cloud_sort.hpp"
#include "common.hpp"
#include "../../include/sn_expf.h"

namespace {

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

// wave-uniform minimum over the 64 lanes
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = row_min_u32(v);
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return umin32(umin32(a, b), umin32(c, d));
}

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
    int *__restrict__ idxs) {
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
        tmp[i] = v + __builtin_ldexpf(e, (int)(low[i] & 1u));
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
    out[e] = feat[bc * n + idx[b * m + j]];
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
    unsafeAtomicAdd(&grad_feat[bc * n + idx[b * m + j]], grad_out[e]);
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

extern "C" size_t sn_mds_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  if (mds_use_clustered(n))  // perm + cell ids + cell offsets + bounding boxes
    return sn::align_up((size_t)b * n * 4, 256) * 2 + (size_t)b * kSortCells * 4 + 256 * (size_t)b;
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
    float *bbox = reinterpret_cast<float *>(w);
    SN_REQUIRE(cloud_sort(b, n, xyz, bbox, hist, cell_of, perm, s) == 0, "sn_mds: cannot size the sort kernel's LDS");
    const size_t lds = (size_t)ppt * 1024 * 8;
#define SN_MDSC(P)                                                                               \
  {                                                                                              \
    /* every call: the attribute belongs to the CURRENT device (several devices per process under    \
       DataParallel), it is not a per-process fact */                                            \
    SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_clustered_kernel<P>),         \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));  \
    mds_clustered_kernel<P><<<b, 1024, lds, s>>>(n, m, xyz, perm, bbox, mean_mst_length, idx);          \
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
