// expansion.hip -- expansion penalty (per-patch MST) for MI355X (gfx950).
//
// Reference: cuda/expansion_penalty/expansion_penalty_cuda.cu:7-149 (forward),
// :167-184 (backward).  Semantics (oracle/expansion.c is the sequential
// statement): Prim from vertex 0 on sqrtf lengths, next vertex = arg-min with the
// HIGHEST index winning ties, strict '<' relaxations; mean edge length through a
// balanced pairwise sum; leaf stripping with snapshot reads; an edge longer than
// alpha*mean is charged to the endpoint that was stripped first.
//
// MI355X design: ONE WAVE PER PATCH.  A patch is at most 512 points = 8 points
// per lane, so the whole Prim state (cur_dis, cur_idx, visited mask) lives in
// VGPRs, the arg-min is a 6-step wave64 xor-butterfly and there is NO barrier in
// the 511 sequential rounds (the reference runs ~11 __syncthreads per round on a
// 512-thread block).  The reference's two [B, n*512] neighbor/cost scratch
// tensors (2 x 1.07 GB at B=32, n=16384) are replaced by the parent/weight of
// each vertex (a tree has P-1 edges): 4 KB of LDS.  Leaf stripping is edge
// centric: the lane that owns child c decides the edge (c, parent[c]) from a
// snapshot of both endpoint degrees.
#include "common.hpp"
#include "wave_dpp.hpp"

namespace {

constexpr int kPMax = 512;

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }

// wave-uniform minimum over the 64 lanes: wave_dpp.hpp (one definition for the sampler, the expansion penalty and
// the renderer)
using sn::wave_min_u32;

// A vertex that is in the tree (or a padding slot) carries this key: as a SIGNED int it is below every
// squared length, so the relaxation test never fires; as an UNSIGNED int it is above every squared
// length, so the arg-min never returns it.
constexpr unsigned kInTree = 0x80000000u;
// rn(sqrt(a)) == rn(sqrt(b)) with a < b needs b - a <= 4 ulp(a): sqrt(b) - sqrt(a) <= ulp(s) <= 2^-23 s,
// times sqrt(a) + sqrt(b) <= 2 sqrt(b), is <= 2^-22 b, and ulp(a) > 2^-24 a.  Twice that is the
// window inside which a decision on squared lengths is re-taken on the rounded square roots.
constexpr unsigned kSqrtTieUlps = 8u;

template <int PPL>
__global__ __launch_bounds__(64) void expansion_fwd_kernel(
    int n, int P, const float *__restrict__ xyz, float alpha, float *__restrict__ dist,
    int *__restrict__ assignment, float *__restrict__ patch_mean) {
#pragma clang fp contract(off)
  __shared__ float4 pts[kPMax];
  __shared__ int s_cnt[kPMax];
  __shared__ float s_dist[kPMax];
  __shared__ int s_assign[kPMax];

  const int b = blockIdx.y, patch = blockIdx.x, lane = threadIdx.x;
  const int base = patch * P;
  const float *__restrict__ src = xyz + ((size_t)b * n + base) * 3;
  const int L = P / PPL;  // active lanes (P < 64 => PPL == 1, L == P)

  for (int v = lane; v < P; v += 64) {
    pts[v] = make_float4(src[v * 3 + 0], src[v * 3 + 1], src[v * 3 + 2], 0.f);
    s_cnt[v] = 0;
    s_dist[v] = 0.f;
    s_assign[v] = -1;
  }
  __syncthreads();

  // Prim on SQUARED lengths held as their bit patterns (monotone for non-negative floats).  The
  // reference orders by sqrtf(d2); rn(sqrt) is monotone, so a decision taken on d2 is the reference's
  // decision unless the two d2 are within kSqrtTieUlps of each other -- those rounds (exact ties on
  // lattice data, one in ~1e4 rounds on random data) are re-decided on the square roots themselves.
  constexpr int NP = (PPL + 1) / 2;
  const bool lane_on = lane < L;
  f2 px[NP], py[NP], pz[NP];
  unsigned key[2 * NP];
  int par[2 * NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int v0 = lane_on ? lane * PPL + 2 * i : 0, v1 = (lane_on && 2 * i + 1 < PPL) ? v0 + 1 : v0;
    const float4 q0 = pts[v0], q1 = pts[v1];
    px[i] = f2{q0.x, q1.x};
    py[i] = f2{q0.y, q1.y};
    pz[i] = f2{q0.z, q1.z};
    key[2 * i] = lane_on ? __float_as_uint(1e18f) : kInTree;
    key[2 * i + 1] = (lane_on && 2 * i + 1 < PPL) ? __float_as_uint(1e18f) : kInTree;
    par[2 * i] = par[2 * i + 1] = 0;
  }
  if (lane == 0) key[0] = kInTree;
  int last = 0;

  for (int round = 0; round < P - 1; ++round) {
    const float4 ql = pts[last];  // wave-uniform LDS read
    const f2 qx = f2{ql.x, ql.x}, qy = f2{ql.y, ql.y}, qz = f2{ql.z, ql.z};
    unsigned d2[2 * NP], t[2 * NP], nearest = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const f2 dx = px[i] - qx, dy = py[i] - qy, dz = pz[i] - qz;
      const f2 d = (dx * dx + dy * dy) + dz * dz;
      d2[2 * i] = __float_as_uint(d.x);
      d2[2 * i + 1] = __float_as_uint(d.y);
    }
#pragma unroll
    for (int r = 0; r < 2 * NP; ++r) {
      t[r] = key[r] - d2[r];
      nearest = umin32(nearest, t[r]);
    }
    if (__builtin_expect(__any(nearest <= kSqrtTieUlps), 0)) {
#pragma unroll
      for (int r = 0; r < 2 * NP; ++r) {
        bool relax = (int)d2[r] < (int)key[r];
        if (relax && t[r] <= kSqrtTieUlps)
          relax = __builtin_sqrtf(__uint_as_float(d2[r])) < __builtin_sqrtf(__uint_as_float(key[r]));
        key[r] = relax ? d2[r] : key[r];
        par[r] = relax ? last : par[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 2 * NP; ++r) {
        const bool relax = (int)d2[r] < (int)key[r];
        key[r] = relax ? d2[r] : key[r];
        par[r] = relax ? last : par[r];
      }
    }

    // next vertex: the smallest sqrt length, highest index among equals
    unsigned lmin = key[0];
#pragma unroll
    for (int r = 1; r < 2 * NP; ++r) lmin = umin32(lmin, key[r]);
    const unsigned thr = wave_min_u32(lmin) + kSqrtTieUlps;
    unsigned long long hit[2 * NP], any = 0;
    int total = 0;
#pragma unroll
    for (int r = 0; r < 2 * NP; ++r) {
      hit[r] = __ballot(key[r] <= thr);
      total += __popcll(hit[r]);
      any |= hit[r];
    }
    if (__builtin_expect(total == 1, 1)) {
      int mine = 0;
#pragma unroll
      for (int r = 0; r < 2 * NP; ++r) {
        const bool h = key[r] <= thr;
        mine = h ? lane * PPL + r : mine;
        key[r] = h ? kInTree : key[r];
      }
      last = __builtin_amdgcn_readlane(mine, __builtin_ctzll(any));
    } else {
      float bd = 1e9f;
      int bv = lane_on ? lane * PPL : -1;  // in-tree / idle lanes carry (1e9, their slot)
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
        const float cand = key[r] == kInTree ? 1e9f : __builtin_sqrtf(__uint_as_float(key[r]));
        // ascending r with '<=' keeps the highest index among equal minima
        if (lane_on && cand <= bd) {
          bd = cand;
          bv = lane * PPL + r;
        }
      }
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        const float od = __shfl_xor(bd, m);
        const int ov = __shfl_xor(bv, m);
        const bool take = (od < bd) || (od == bd && ov > bv);
        bd = take ? od : bd;
        bv = take ? ov : bv;
      }
      last = bv;
#pragma unroll
      for (int r = 0; r < PPL; ++r)
        if (lane * PPL + r == last) key[r] = kInTree;
    }
  }

  // ---- parent edge of every vertex (the root and anything never reached have none); the length is
  // recomputed with the operands and the operation order of the relaxation that set the parent
  float wgt[PPL];
  unsigned alive = 0;
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const int v = lane * PPL + r;
    const bool has = lane_on && v != 0 && key[r] == kInTree;
    par[r] = has ? par[r] : -1;
    const float4 q = pts[has ? par[r] : 0];
    const float x = (r & 1) ? px[r / 2].y : px[r / 2].x, y = (r & 1) ? py[r / 2].y : py[r / 2].x,
                z = (r & 1) ? pz[r / 2].y : pz[r / 2].x;
    const float dx = x - q.x, dy = y - q.y, dz = z - q.z;
    wgt[r] = has ? __builtin_sqrtf((dx * dx + dy * dy) + dz * dz) : 0.f;
    if (has) {
      alive |= 1u << r;
      atomicAdd(&s_cnt[v], 1);
      atomicAdd(&s_cnt[par[r]], 1);
    }
  }

  // ---- mean edge length: balanced pairwise tree over vertex order
  float acc[PPL];
#pragma unroll
  for (int r = 0; r < PPL; ++r) acc[r] = wgt[r];
#pragma unroll
  for (int s = 1; s < PPL; s <<= 1)
#pragma unroll
    for (int r = 0; r < PPL; r += 2 * s) acc[r] = acc[r] + acc[r + s];
  float total = acc[0];
  for (int m = 1; m < L; m <<= 1) total = total + __shfl_xor(total, m);
  const float mean_dis = total / (float)(P - 1);
  if (lane == 0) patch_mean[(size_t)b * gridDim.x + patch] = mean_dis;
  const float thr = mean_dis * alpha;
  __syncthreads();

  // ---- leaf stripping, snapshot semantics
  for (;;) {
    int sc[PPL], sp[PPL];
    bool leaf = false;
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int v = lane * PPL + r;
      sc[r] = lane_on ? s_cnt[v] : 0;
      sp[r] = ((alive >> r) & 1u) ? s_cnt[par[r]] : 0;
      leaf = leaf || (sc[r] == 1);
    }
    if (!__any(leaf)) break;
    __syncthreads();  // every snapshot read precedes every decrement
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      if (!((alive >> r) & 1u)) continue;
      const int c = lane * PPL + r, p = par[r];
      int owner = -1, other = -1;
      if (sc[r] == 1 && (sp[r] > 1 || (sp[r] == 1 && c > p))) {
        owner = c;
        other = p;
      } else if (sp[r] == 1 && (sc[r] > 1 || (sc[r] == 1 && p > c))) {
        owner = p;
        other = c;
      }
      if (owner >= 0) {
        alive &= ~(1u << r);
        atomicSub(&s_cnt[c], 1);
        atomicSub(&s_cnt[p], 1);
        if (wgt[r] > thr) {
          s_dist[owner] = wgt[r];
          s_assign[owner] = base + other;
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  for (int v = lane; v < P; v += 64) {
    dist[(size_t)b * n + base + v] = s_dist[v];
    assignment[(size_t)b * n + base + v] = s_assign[v];
  }
}

// (Round 6 built a TWO-WAVES-PER-PATCH form for the 4 / 8 / 16-cloud shares of a multi-GPU job -- 128 lanes own a patch,
// half the per-round vector work per wave, one LDS exchange {key, count, vertex, x, y, z} + one workgroup barrier per
// round, the square-root tie branch across the two waves: bit-identical on every test, and NOT faster: 0.335 against
// 0.325 ms at 4 clouds, 0.372 against 0.360 at 32 (profiles/r06_d_expansion_two_waves_not_kept.txt).  A round is a
// chain of ~150 DEPENDENT instructions of one wave (LDS read -> distances -> relaxation -> wave minimum -> ballots ->
// readlane), 640 ns; halving the lanes' work shortens none of its links and the exchange adds one.  Removed.)

// mean_mst_length[b] = sum over patches in ascending patch order (fixed order:
// the reference's fp32 atomicAdd is order dependent)
__global__ void expansion_mean_kernel(int B, int np, const float *__restrict__ patch_mean,
                                      float *__restrict__ mean_mst_length) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int p = 0; p < np; ++p) acc += patch_mean[(size_t)b * np + p];
  mean_mst_length[b] = acc;
}

__global__ __launch_bounds__(256) void expansion_bwd_kernel(
    int B, int n, const float *__restrict__ xyz, const float *__restrict__ graddist,
    const int *__restrict__ assignment, float *__restrict__ grad) {
  const long total = (long)B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int j2 = assignment[e];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (j2 != -1) {
      const long bb = e / n;
      const float *a = xyz + e * 3, *o = xyz + (bb * n + j2) * 3;
      const float g = graddist[e] * 2;
      g0 = g * (a[0] - o[0]);
      g1 = g * (a[1] - o[1]);
      g2 = g * (a[2] - o[2]);
    }
    grad[e * 3 + 0] = g0;
    grad[e * 3 + 1] = g1;
    grad[e * 3 + 2] = g2;
  }
}

}  // namespace

extern "C" size_t sn_expansion_workspace_bytes(int b, int n, int primitive_size) {
  if (b < 1 || n < 1 || primitive_size < 1) return 0;
  return sn::align_up((size_t)b * (n / primitive_size) * 4, 256);
}

extern "C" int sn_expansion_forward(const float *xyz, int b, int n, int primitive_size,
                                    float alpha, float *dist, int *assignment,
                                    float *mean_mst_length, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  SN_REQUIRE(xyz && dist && assignment && mean_mst_length && workspace,
             "sn_expansion_forward: null pointer");
  const int P = primitive_size;
  SN_REQUIRE(b >= 1 && n >= 1, "sn_expansion_forward: need b,n >= 1");
  SN_REQUIRE(P >= 2 && P <= 512 && (P & (P - 1)) == 0,
             "sn_expansion_forward: primitive_size must be a power of two in [2,512] (got %d)", P);
  SN_REQUIRE(n % P == 0, "sn_expansion_forward: n (%d) must be a multiple of primitive_size (%d)", n, P);
  SN_REQUIRE(b <= 65535, "sn_expansion_forward: batch too large");
  SN_REQUIRE(workspace_bytes >= sn_expansion_workspace_bytes(b, n, P),
             "sn_expansion_forward: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  float *patch_mean = static_cast<float *>(workspace);
  const dim3 grid(n / P, b);
  if (sn::prof_enabled()) sn::prof_begin("expansion_fwd", s);
  switch (P <= 64 ? 1 : P / 64) {
    case 1: expansion_fwd_kernel<1><<<grid, 64, 0, s>>>(n, P, xyz, alpha, dist, assignment, patch_mean); break;
    case 2: expansion_fwd_kernel<2><<<grid, 64, 0, s>>>(n, P, xyz, alpha, dist, assignment, patch_mean); break;
    case 4: expansion_fwd_kernel<4><<<grid, 64, 0, s>>>(n, P, xyz, alpha, dist, assignment, patch_mean); break;
    default: expansion_fwd_kernel<8><<<grid, 64, 0, s>>>(n, P, xyz, alpha, dist, assignment, patch_mean); break;
  }
  if (sn::prof_enabled()) sn::prof_end("expansion_fwd", s);
  expansion_mean_kernel<<<sn::ceil_div(b, 64), 64, 0, s>>>(b, n / P, patch_mean, mean_mst_length);
  return sn::launch_status("sn_expansion_forward");
}

extern "C" int sn_expansion_backward(const float *xyz, const float *graddist,
                                     const int *assignment, int b, int n, float *gradxyz,
                                     void *stream) {
  SN_REQUIRE(xyz && graddist && assignment && gradxyz, "sn_expansion_backward: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1, "sn_expansion_backward: need b,n >= 1");
  const long total = (long)b * n;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  expansion_bwd_kernel<<<blocks, 256, 0, sn::as_stream(stream)>>>(b, n, xyz, graddist,
                                                                  assignment, gradxyz);
  return sn::launch_status("sn_expansion_backward");
}
