// knn_mfma.hip -- EdgeConv's k-nearest-neighbour search as ONE kernel on the fp32 matrix cores (gfx950).
//
// Reference: knn() of models/sparenet_generator.py:852-877: pairwise = -|x_i|^2 + 2 x_i.x_j - |x_j|^2,
// idx = pairwise.topk(k)  (GPU branch: the un-vendored KNN_CUDA wheel).  This is the one place of the
// hot path whose search IS a dense contraction (K = C channels, up to 512).  sn_knn_topk (knn.hip)
// ranks a score matrix that a library GEMM wrote to HBM: 1.15 GB written and read back at B = 32,
// N = 3000.  Here the score tile never leaves the registers:
//
//   workgroup = 128 queries x all keys of one cloud, 4 waves as 2 (query halves) x 2 (key halves);
//   per step of 128 keys a wave owns a 64 keys x 64 queries block = 2 x 2 accumulators of
//   v_mfma_f32_32x32x2_f32 (A = 32 keys x 2 channels, B = 2 channels x 32 queries, 16 registers each);
//   x is [C][N] in memory, which is exactly the operand layout: lane l of an A (B) operand reads channel
//   2 s + (l >> 5) of key (query) l & 31 -- unit stride along the points, no transpose anywhere;
//   channels move through LDS in chunks of 16 (two 8 KB tiles, double buffered, one barrier per chunk);
//   the accumulators start at -|x_j|^2 and the queries are stored doubled, so a finished accumulator is
//   the ranking score 2 x_i.x_j - |x_j|^2 (the row constant -|x_i|^2 does not change the order);
//   in the 32x32 result layout a lane holds ONE query column and 16 key rows per accumulator: it keeps
//   the k best of its two queries in registers (sorted insertion, ascending key order, so equal scores
//   keep the lower index) -- no cross-lane traffic in the scan;
//   at the end the 4 partial lists of a query (2 row halves x 2 key halves) meet in LDS.
// The point itself is forced to the front (in exact arithmetic it is the unique maximum).
// fp32 throughout: 2 N^2 C flop per cloud on the 157 TFLOP/s fp32 MFMA peak.
#include "common.hpp"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kTile = 128;   // queries per workgroup, keys per step
constexpr int kChunk = 16;   // channels per LDS stage

__global__ __launch_bounds__(256) void knn_sqnorm_kernel(const float *__restrict__ x, int c, int n,
                                                         long total, float *__restrict__ xx) {
#pragma clang fp contract(off)
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / n;
    const int j = (int)(e - b * n);
    const float *p = x + b * c * n + j;
    float s = 0.f;
    for (int ch = 0; ch < c; ++ch) s += p[(size_t)ch * n] * p[(size_t)ch * n];
    xx[e] = s;
  }
}

// one tile row of 128 points of channel `ch` starting at point p0 -> 4 consecutive floats of this thread
__device__ __noinline__ f4v load_row4(const float *__restrict__ xb, int c, int n, int ch, int p, bool vec) {
  f4v v = {0.f, 0.f, 0.f, 0.f};
  if (ch < c) {
    const float *src = xb + (size_t)ch * n + p;
    if (vec && p + 3 < n) {
      v = *reinterpret_cast<const f4v *>(src);
    } else {
      if (p + 0 < n) v.x = src[0];
      if (p + 1 < n) v.y = src[1];
      if (p + 2 < n) v.z = src[2];
      if (p + 3 < n) v.w = src[3];
    }
  }
  return v;
}

// K = 8: three workgroups per CU (<= 168 registers): the generator's 32 clouds x 24 strips of 128 queries
// are exactly 3 x 256 workgroups, one round of the chip instead of 1.5
template <int K>
__global__ __launch_bounds__(256, K <= 8 ? 3 : 1) void knn_mfma_kernel(const float *__restrict__ x,
                                                       const float *__restrict__ xx, int c, int n, int k,
                                                       long long *__restrict__ idx) {
  extern __shared__ float smem[];
  // [buffer 2][operand 2 (0 = keys, 1 = queries)][kChunk][kTile], then the key norms of the step
  float *tiles = smem;
  float *xxs = smem + 2 * 2 * kChunk * kTile;
  const int b = blockIdx.y, q0 = blockIdx.x * kTile;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wq = wave & 1, wk = wave >> 1;
  const int h = lane >> 5, c31 = lane & 31;
  const float *xb = x + (size_t)b * c * n;
  const float *xxb = xx + (size_t)b * n;
  const bool vec = (n & 3) == 0;  // rows start 16-byte aligned

  float lv[2][K];
  int li[2][K];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < K; ++i) {
      lv[qb][i] = -3.0e38f;
      li[qb][i] = 0x7fffffff;
    }

  // loader: thread t stages rows (t >> 5) and (t >> 5) + 8 of both operands, 4 points each
  const int lrow = tid >> 5, lcol = (tid & 31) * 4;
  const int nchunks = (c + kChunk - 1) / kChunk;
  const int ntiles = (n + kTile - 1) / kTile;

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * kTile;
    if (tid < kTile) xxs[tid] = k0 + tid < n ? xxb[k0 + tid] : 0.f;
    f4v stage[4];
    // interior steps (full 128-point tiles, 16-byte aligned rows) load without a single test
    const bool interior = vec && k0 + kTile <= n && q0 + kTile <= n;
    auto fetch = [&](int ch0) {
      if (interior && ch0 + kChunk <= c) {
        const float *r0 = xb + (size_t)(ch0 + lrow) * n + lcol, *r1 = r0 + (size_t)8 * n;
        stage[0] = *reinterpret_cast<const f4v *>(r0 + k0);
        stage[1] = *reinterpret_cast<const f4v *>(r1 + k0);
        stage[2] = *reinterpret_cast<const f4v *>(r0 + q0);
        stage[3] = *reinterpret_cast<const f4v *>(r1 + q0);
      } else {
        stage[0] = load_row4(xb, c, n, ch0 + lrow, k0 + lcol, vec);
        stage[1] = load_row4(xb, c, n, ch0 + lrow + 8, k0 + lcol, vec);
        stage[2] = load_row4(xb, c, n, ch0 + lrow, q0 + lcol, vec);
        stage[3] = load_row4(xb, c, n, ch0 + lrow + 8, q0 + lcol, vec);
      }
    };
    auto commit = [&](int buf) {
      float *kt_ = tiles + (size_t)buf * 2 * kChunk * kTile;
      float *qt_ = kt_ + kChunk * kTile;
      *reinterpret_cast<f4v *>(kt_ + lrow * kTile + lcol) = stage[0];
      *reinterpret_cast<f4v *>(kt_ + (lrow + 8) * kTile + lcol) = stage[1];
      *reinterpret_cast<f4v *>(qt_ + lrow * kTile + lcol) = stage[2] * 2.0f;  // exact
      *reinterpret_cast<f4v *>(qt_ + (lrow + 8) * kTile + lcol) = stage[3] * 2.0f;
    };
    fetch(0);
    commit(0);
    __syncthreads();

    f16v acc[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f16v init;
#pragma unroll
      for (int a = 0; a < 4; ++a) {  // result rows 8 a + 4 h + (0..3) of this key block
        const f4v nx = *reinterpret_cast<const f4v *>(xxs + wk * 64 + kb * 32 + 8 * a + 4 * h);
        init[4 * a + 0] = -nx.x;
        init[4 * a + 1] = -nx.y;
        init[4 * a + 2] = -nx.z;
        init[4 * a + 3] = -nx.w;
      }
      acc[kb][0] = init;
      acc[kb][1] = init;
    }

    for (int ch = 0; ch < nchunks; ++ch) {
      if (ch + 1 < nchunks) fetch((ch + 1) * kChunk);
      const float *kt_ = tiles + (size_t)(ch & 1) * 2 * kChunk * kTile + wk * 64 + c31;
      const float *qt_ = tiles + (size_t)(ch & 1) * 2 * kChunk * kTile + kChunk * kTile + wq * 64 + c31;
#pragma unroll
      for (int s = 0; s < kChunk / 2; ++s) {
        const float a0 = kt_[(2 * s + h) * kTile], a1 = kt_[(2 * s + h) * kTile + 32];
        const float b0 = qt_[(2 * s + h) * kTile], b1 = qt_[(2 * s + h) * kTile + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      if (ch + 1 < nchunks) commit((ch + 1) & 1);
      __syncthreads();
    }

    // The lane's two queries take their 32 scores of this step in ascending key order.  Almost all of
    // them lose against the lane's current k-th best: each lane marks its own candidates in a 32-bit
    // mask while the scores pass into the (now idle) operand tiles, then the wave loops as long as any
    // lane has a candidate left -- 2-4 trips of ONE run-time indexed insertion instead of 32.
    float *scratch = tiles + (size_t)wave * 32 * 64 + lane;
    const bool own_tile = kt == (int)blockIdx.x;    // holds the queries themselves
    const bool last_tile = k0 + kTile > n;          // holds keys past the end
    const int kbase = k0 + wk * 64 + 4 * h;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = q0 + wq * 64 + qb * 32 + c31;
      const float thr = lv[qb][K - 1];
      unsigned cand = 0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[kb][qb][r];
          if (own_tile || last_tile) {  // wave-uniform, one or two steps per workgroup
            const int key = kbase + kb * 32 + 8 * (r >> 2) + (r & 3);
            v = key == qi ? 3.0e38f : v;  // the point itself first
            v = key < n ? v : -3.0e38f;
          }
          scratch[(kb * 16 + r) * 64] = v;
          cand |= v > thr ? 1u << (kb * 16 + r) : 0u;
        }
      while (__any(cand != 0u)) {
        if (cand != 0u) {
          const int r = __builtin_ctz(cand);
          cand &= cand - 1;
          const int key = kbase + (r >> 4) * 32 + 8 * ((r >> 2) & 3) + (r & 3);
          const float v = scratch[r * 64];
          if (v > lv[qb][K - 1]) {  // the threshold may have risen since the mask was taken
#pragma unroll
            for (int p = K - 1; p > 0; --p) {
              const bool up = v > lv[qb][p - 1];  // the slot above moves down
              const bool here = v > lv[qb][p];
              lv[qb][p] = up ? lv[qb][p - 1] : (here ? v : lv[qb][p]);
              li[qb][p] = up ? li[qb][p - 1] : (here ? key : li[qb][p]);
            }
            const bool top = v > lv[qb][0];
            lv[qb][0] = top ? v : lv[qb][0];
            li[qb][0] = top ? key : li[qb][0];
          }
        }
      }
    }
    __syncthreads();  // the tiles are operands again
  }

  // ---- merge the four partial lists of every query (sources: 2 key halves x 2 row halves)
  __syncthreads();
  float *mv = smem;                                             // [kTile][4][K]
  int *mi = reinterpret_cast<int *>(smem + kTile * 4 * K);      // [kTile][4][K]
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int ql = wq * 64 + qb * 32 + c31;
    const int src = wk * 2 + h;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      mv[(ql * 4 + src) * K + i] = lv[qb][i];
      mi[(ql * 4 + src) * K + i] = li[qb][i];
    }
  }
  __syncthreads();
  if (tid < kTile && q0 + tid < n) {
    int head[4] = {0, 0, 0, 0};
    long long *out = idx + ((size_t)b * n + q0 + tid) * k;
    for (int j = 0; j < k; ++j) {
      float bv = -3.0e38f;
      int bi = 0x7fffffff, bs = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool has = head[s] < K;
        const float v = has ? mv[(tid * 4 + s) * K + head[s]] : -3.0e38f;
        const int id = has ? mi[(tid * 4 + s) * K + head[s]] : 0x7fffffff;
        if (v > bv || (v == bv && id < bi)) {
          bv = v;
          bi = id;
          bs = s;
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) head[s] += s == bs ? 1 : 0;
      out[j] = bi;
    }
  }
}

template <int K>
size_t knn_lds_bytes() {
  const size_t tiles = (size_t)(2 * 2 * kChunk * kTile + kTile) * 4;
  const size_t merge = (size_t)kTile * 4 * K * 8;
  return tiles > merge ? tiles : merge;
}

template <int K>
int launch_knn(const float *x, const float *xx, int b, int c, int n, int k, long long *idx, hipStream_t s) {
  const size_t lds = knn_lds_bytes<K>();
  if (lds > 48 * 1024)  // per call: the attribute belongs to the current device
    SN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_mfma_kernel<K>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((n + kTile - 1) / kTile, b);
  knn_mfma_kernel<K><<<grid, 256, lds, s>>>(x, xx, c, n, k, idx);
  return sn::launch_status("sn_knn");
}

}  // namespace

extern "C" size_t sn_knn_workspace_bytes(int b, int n) {
  if (b < 1 || n < 1) return 0;
  return sn::align_up((size_t)b * n * 4, 256);
}

extern "C" int sn_knn(const float *x, int b, int c, int n, int k, long long *idx, void *workspace,
                      size_t workspace_bytes, void *stream) {
  SN_REQUIRE(x && idx && workspace, "sn_knn: null pointer");
  SN_REQUIRE(b >= 1 && c >= 1 && n >= 1, "sn_knn: need b, c, n >= 1");
  SN_REQUIRE(k >= 1 && k <= 20 && k <= n, "sn_knn: need 1 <= k <= min(n, 20) (got %d)", k);
  SN_REQUIRE(b <= 65535, "sn_knn: batch too large");
  SN_REQUIRE(workspace_bytes >= sn_knn_workspace_bytes(b, n), "sn_knn: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  float *xx = static_cast<float *>(workspace);
  const long total = (long)b * n;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  knn_sqnorm_kernel<<<blocks, 256, 0, s>>>(x, c, n, total, xx);
  if (k <= 8) return launch_knn<8>(x, xx, b, c, n, k, idx, s);
  return launch_knn<20>(x, xx, b, c, n, k, idx, s);
}
