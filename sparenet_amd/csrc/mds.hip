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

int lin_blocks(long total) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t sn_mds_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
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
    static bool once = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_kernel<20, 2, 1024>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) ==
             hipSuccess;
    }();
    (void)once;
    SN_MDS_Z(20, 2);
  } else if (ppt <= 24) {
    static bool once = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void *>(&mds_kernel<24, 1, 1024>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) ==
             hipSuccess;
    }();
    (void)once;
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
