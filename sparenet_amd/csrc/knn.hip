// knn.hip -- EdgeConv's k-nearest-neighbour graph in feature space and its edge features (gfx950).
//
// Reference: models/sparenet_generator.py:852-877 (knn), :880-906 (get_graph_feature).  On the
// GPU the reference calls the un-vendored wheel KNN_CUDA 0.2 (github.com/unlimblue/KNN_CUDA: all
// N x N squared distances, per-column insertion sort of the k smallest, ascending, the point
// itself first); its CPU branch ranks  -|x_i|^2 + 2 x_i.x_j - |x_j|^2  with torch.topk.
// Split here the way the work splits on this chip:
//   * the N x N inner products are one plain batched GEMM (x^T x, K = C up to 512) -- left to
//     rocBLAS through torch.bmm on the host side;
//   * sn_knn_topk: one wave per row ranks  s_j = |x_j|^2 - 2 x_i.x_j  (the row constant |x_i|^2
//     does not change the order), every lane keeps the k smallest of its strided share in
//     registers, then k wave-minimum extractions merge them; equal scores resolve to the lower
//     index.  HBM bound: the matrix is read once (1.15 GB at B=32, N=3000).
//   * sn_graph_feature_forward / backward: out[b, c, n, j] = x[b, c, idx[b,n,j]] - x[b, c, n],
//     out[b, C + c, n, j] = x[b, c, n]  (the cat((feature - x, x)).permute(0,3,1,2) of :899-905).
#include "common.hpp"

namespace {

constexpr int kMaxK = 32;

template <int K>
__global__ __launch_bounds__(256) void knn_topk_kernel(const float *__restrict__ inner,
                                                       const float *__restrict__ xx, int n, long rows,
                                                       long long *__restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long b = row / n;
  const float *in = inner + row * n;
  const float *x2 = xx + b * n;
  // the lane's K smallest (score, index), ascending; (3e38, 2^31-1) = empty
  float v[K];
  int id[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    v[i] = 3.0e38f;
    id[i] = 0x7fffffff;
  }
  for (int j = lane; j < n; j += 64) {
    const float s = __builtin_fmaf(-2.f, in[j], x2[j]);
    if (s < v[K - 1]) {  // strided ascending j: an equal score never displaces an earlier index
      float cv = s;
      int ci = j;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const bool lt = cv < v[i];
        const float tv = v[i];
        const int ti = id[i];
        v[i] = lt ? cv : tv;
        id[i] = lt ? ci : ti;
        cv = lt ? tv : cv;
        ci = lt ? ti : ci;
      }
    }
  }
  // K extractions of the wave-wide minimum (score, index); the owner pops its head
  for (int r = 0; r < K; ++r) {
    float mv = v[0];
    int mi = id[0];
    for (int m = 1; m < 64; m <<= 1) {
      const float ov = __shfl_xor(mv, m);
      const int oi = __shfl_xor(mi, m);
      const bool take = ov < mv || (ov == mv && oi < mi);
      mv = take ? ov : mv;
      mi = take ? oi : mi;
    }
    if (lane == 0) idx[row * K + r] = mi == 0x7fffffff ? 0 : mi;
    if (id[0] == mi && v[0] == mv) {  // indices are unique: exactly one lane
#pragma unroll
      for (int i = 0; i + 1 < K; ++i) {
        v[i] = v[i + 1];
        id[i] = id[i + 1];
      }
      v[K - 1] = 3.0e38f;
      id[K - 1] = 0x7fffffff;
    }
  }
}

__global__ __launch_bounds__(256) void graph_feature_fwd_kernel(const float *__restrict__ x,
                                                                const long long *__restrict__ idx,
                                                                int c, int n, int k, long total,
                                                                float *__restrict__ out) {
  // one thread per (b, ch, n, j); out [B, 2C, N, k]
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e % k);
    const long t = e / k;
    const int p = (int)(t % n);
    const long t2 = t / n;
    const int ch = (int)(t2 % c);
    const long b = t2 / c;
    const float *xb = x + (b * c + ch) * n;
    const float self = xb[p];
    const float nb = xb[idx[(b * n + p) * k + j]];
    out[((b * 2 * c + ch) * n + p) * k + j] = nb - self;
    out[((b * 2 * c + c + ch) * n + p) * k + j] = self;
  }
}

// Backward without floating-point atomics: the graph is the same for all channels, so the edges
// are first inverted once (for every point q the list of edges (p, j) with idx[p, j] == q: count,
// scan, fill), then one thread per (b, ch, q) adds its own terms and the terms of its incoming edges.
__global__ __launch_bounds__(256) void graph_count_kernel(const long long *__restrict__ idx, int n,
                                                          int k, long edges, int *__restrict__ cnt) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < edges; e += (long)gridDim.x * blockDim.x) {
    const long b = e / ((long)n * k);
    atomicAdd(&cnt[b * n + idx[e]], 1);
  }
}

// per cloud: in-place exclusive scan of the n counts
__global__ __launch_bounds__(1024) void graph_scan_kernel(int *__restrict__ cnt, int n) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x;
  int *c = cnt + (size_t)blockIdx.x * n;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? c[i] : 0;
    int incl = v;
    for (int m = 1; m < 64; m <<= 1) {
      const int u = __shfl_up(incl, m);
      if ((tid & 63) >= m) incl += u;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < (tid >> 6); ++w) pre += wsum[w];
    if (i < n) c[i] = pre + incl - v;
    __syncthreads();
    if (tid == 1023) carry = pre + incl;
    __syncthreads();
  }
}

// elist[b][start(q) ...] = edge ids p * k + j; afterwards offs[b][q] is the END of q's list
__global__ __launch_bounds__(256) void graph_fill_kernel(const long long *__restrict__ idx, int n, int k,
                                                         long edges, int *__restrict__ offs,
                                                         int *__restrict__ elist) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < edges; e += (long)gridDim.x * blockDim.x) {
    const long b = e / ((long)n * k);
    const int local = (int)(e - b * n * k);
    const int pos = atomicAdd(&offs[b * n + idx[e]], 1);
    elist[b * (long)n * k + pos] = local;
  }
}

__global__ __launch_bounds__(256) void graph_feature_bwd_kernel(const float *__restrict__ g,
                                                                const int *__restrict__ offs,
                                                                const int *__restrict__ elist, int c,
                                                                int n, int k, long total,
                                                                float *__restrict__ gx) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e % n);
    const long t = e / n;
    const int ch = (int)(t % c);
    const long b = t / c;
    const float *g1 = g + (b * 2 * c + ch) * (long)n * k;       // d out / d (neighbour - point)
    const float *g2 = g + (b * 2 * c + c + ch) * (long)n * k;   // d out / d point
    float acc = 0.f;
    for (int j = 0; j < k; ++j) acc += g2[(long)q * k + j] - g1[(long)q * k + j];
    const int beg = q > 0 ? offs[b * n + q - 1] : 0, end = offs[b * n + q];
    const int *el = elist + b * (long)n * k;
    for (int i = beg; i < end; ++i) acc += g1[el[i]];
    gx[e] = acc;
  }
}

int blocks_for(long total) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

}  // namespace

extern "C" int sn_knn_topk(const float *inner, const float *xx, int b, int n, int k,
                           long long *idx, void *stream) {
  SN_REQUIRE(inner && xx && idx, "sn_knn_topk: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1 && k >= 1 && k <= kMaxK && k <= n, "sn_knn_topk: need 1 <= k <= min(n, %d)", kMaxK);
  const long rows = (long)b * n;
  SN_REQUIRE((rows + 3) / 4 < (1L << 31), "sn_knn_topk: too many rows");
  const int grid = (int)((rows + 3) / 4);
  hipStream_t s = sn::as_stream(stream);
#define SN_KNN(K) knn_topk_kernel<K><<<grid, 256, 0, s>>>(inner, xx, n, rows, idx)
  switch (k) {
    case 1: SN_KNN(1); break;
    case 2: SN_KNN(2); break;
    case 4: SN_KNN(4); break;
    case 8: SN_KNN(8); break;
    case 16: SN_KNN(16); break;
    case 20: SN_KNN(20); break;
    case 32: SN_KNN(32); break;
    default: return sn::fail(SN_EINVAL, "sn_knn_topk: k must be one of 1, 2, 4, 8, 16, 20, 32 (got %d)", k);
  }
#undef SN_KNN
  return sn::launch_status("sn_knn_topk");
}

extern "C" int sn_graph_feature_forward(const float *x, const long long *idx, int b, int c, int n,
                                        int k, float *out, void *stream) {
  SN_REQUIRE(x && idx && out, "sn_graph_feature_forward: null pointer");
  SN_REQUIRE(b >= 1 && c >= 1 && n >= 1 && k >= 1, "sn_graph_feature_forward: bad sizes");
  const long total = (long)b * c * n * k;
  graph_feature_fwd_kernel<<<blocks_for(total), 256, 0, sn::as_stream(stream)>>>(x, idx, c, n, k, total, out);
  return sn::launch_status("sn_graph_feature_forward");
}

extern "C" size_t sn_graph_feature_backward_workspace_bytes(int b, int n, int k) {
  if (b < 1 || n < 1 || k < 1) return 0;
  return sn::align_up((size_t)b * n * 4, 256) + (size_t)b * n * k * 4;
}

extern "C" int sn_graph_feature_backward(const float *grad_out, const long long *idx, int b, int c,
                                         int n, int k, float *grad_x, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  SN_REQUIRE(grad_out && idx && grad_x && workspace, "sn_graph_feature_backward: null pointer");
  SN_REQUIRE(b >= 1 && c >= 1 && n >= 1 && k >= 1, "sn_graph_feature_backward: bad sizes");
  SN_REQUIRE(workspace_bytes >= sn_graph_feature_backward_workspace_bytes(b, n, k),
             "sn_graph_feature_backward: workspace too small");
  hipStream_t s = sn::as_stream(stream);
  int *offs = static_cast<int *>(workspace);
  int *elist = reinterpret_cast<int *>(static_cast<char *>(workspace) + sn::align_up((size_t)b * n * 4, 256));
  const long edges = (long)b * n * k;
  SN_HIP(hipMemsetAsync(offs, 0, (size_t)b * n * 4, s));
  graph_count_kernel<<<blocks_for(edges), 256, 0, s>>>(idx, n, k, edges, offs);
  graph_scan_kernel<<<b, 1024, 0, s>>>(offs, n);
  graph_fill_kernel<<<blocks_for(edges), 256, 0, s>>>(idx, n, k, edges, offs, elist);
  const long total = (long)b * c * n;
  graph_feature_bwd_kernel<<<blocks_for(total), 256, 0, s>>>(grad_out, offs, elist, c, n, k, total, grad_x);
  return sn::launch_status("sn_graph_feature_backward");
}
