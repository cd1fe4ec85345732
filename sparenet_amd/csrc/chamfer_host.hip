// chamfer_host.hip -- Chamfer distance on HOST tensors (no device code in this file).
//
// The reference's ChamferDistanceFunction takes CPU tensors too (cuda/chamfer_distance/chamfer_distance.py:31-32,
// :53-54 -> cd.forward / cd.backward = chamfer_distance.cpp:57-112 / :114-180, one thread, all pairs); BASELINE config 1
// is that path.  This is the library's own host implementation of the same contract, so that the drop-in accepts
// whatever the reference accepts -- it is NOT a fallback for the GPU ops (CUDA tensors never come here, and every
// other op still refuses CPU tensors).
//
//   forward   dist[b, j] = min_k d(j, k),  d = (dx*dx + dy*dy) + dz*dz in fp32 with dx = t.x - q.x, every product and
//             sum rounded on its own (the library is built with -ffp-contract=off);  idx[b, j] = the LOWEST k attaining
//             the minimum (chamfer_distance.cpp:72-83: "k == 0 || d < best").
//   backward  g = 2 graddist; gradxyz1[j] += g (p1 - p2), gradxyz2[idx1[j]] -= g (p1 - p2) for j = 0 .. n-1, then the
//             symmetric loop over cloud 2 -- the additions of a cloud in exactly the reference's order (:143-178), so
//             the sums are its sums bit for bit (fp32 addition does not commute with reordering).
// Host design: clouds and blocks of 256 queries are dealt to std::threads (the searches are independent); a query
// walks the targets eight interleaved lanes at a time -- lane l sees k = l, l + 8, ... and keeps its own (best, k), so
// the loop vectorises -- and the lanes are merged by (distance, then lower k), which is the sequential rule.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kLanes = 8;
constexpr int kQueryBlock = 256;

void nn_block(const float *q, int q0, int q1, const float *t, int m, float *dist, int *idx) {
  for (int j = q0; j < q1; ++j) {
    const float qx = q[j * 3 + 0], qy = q[j * 3 + 1], qz = q[j * 3 + 2];
    float best[kLanes];
    int bk[kLanes];
    for (int l = 0; l < kLanes; ++l) {
      best[l] = __builtin_inff();
      bk[l] = -1;
    }
    int k = 0;
    for (; k + kLanes <= m; k += kLanes) {
      for (int l = 0; l < kLanes; ++l) {
        const float dx = t[(k + l) * 3 + 0] - qx, dy = t[(k + l) * 3 + 1] - qy, dz = t[(k + l) * 3 + 2] - qz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const bool take = d < best[l];   // strict: the first of equal distances stays; NaN never enters
        best[l] = take ? d : best[l];
        bk[l] = take ? k + l : bk[l];
      }
    }
    float b = __builtin_inff();
    int bi = -1;
    for (int l = 0; l < kLanes; ++l)  // lanes hold disjoint k: smaller distance, then the lower index
      if (bk[l] >= 0 && (bi < 0 || best[l] < b || (best[l] == b && bk[l] < bi))) {
        b = best[l];
        bi = bk[l];
      }
    for (; k < m; ++k) {  // the tail, in order: larger k than everything before
      const float dx = t[k * 3 + 0] - qx, dy = t[k * 3 + 1] - qy, dz = t[k * 3 + 2] - qz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d < b) {
        b = d;
        bi = k;
      }
    }
    {
      // The reference takes target 0 unconditionally and replaces it only on a strictly smaller distance
      // (chamfer_distance.cpp:79): when d(j, 0) is NaN nothing ever replaces it, and when no distance is below +inf
      // target 0 stays as well.
      const float dx = t[0] - qx, dy = t[1] - qy, dz = t[2] - qz;
      const float d0 = (dx * dx + dy * dy) + dz * dz;
      if (d0 != d0 || bi < 0) {
        b = d0;
        bi = 0;
      }
    }
    dist[j] = b;
    idx[j] = bi;
  }
}

// An explicit `threads` argument wins; SN_HOST_THREADS only replaces the default (all hardware threads).  Never more
// threads than items, and few items are not worth a thread each (a thread start costs more than a query block).
int thread_count(int asked, long items) {
  int t = asked;
  if (t <= 0) {
    const char *e = getenv("SN_HOST_THREADS");
    t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (items < 4L * t) t = (int)(items / 4);  // default only: at least four items per thread
  }
  t = t < 1 ? 1 : (t > 256 ? 256 : t);
  return (long)t > items ? (int)items : t;
}

// Items are handed out through a counter, so ANY number of workers finishes the job: a thread that cannot be started
// (std::system_error: EAGAIN under a container's pid limit or inside a DataLoader worker) is simply absent and the
// calling thread takes items itself -- nothing escapes the extern "C" entry points.
template <class F>
void parallel_items(long items, int threads, F fn) {
  if (items <= 0) return;
  const int nt = thread_count(threads, items);
  std::atomic<long> next{0};
  auto work = [&] {
    for (long i = next.fetch_add(1); i < items; i = next.fetch_add(1)) fn(i);
  };
  std::vector<std::thread> pool;
  if (nt > 1) {
    try {
      pool.reserve(nt - 1);
      for (int w = 0; w + 1 < nt; ++w) pool.emplace_back(work);
    } catch (...) {  // run with the threads that did start
    }
  }
  work();
  for (auto &th : pool) th.join();
}

}  // namespace

extern "C" int sn_chamfer_forward_host(const float *xyz1, const float *xyz2, int b, int n, int m, float *dist1,
                                       int *idx1, float *dist2, int *idx2, int threads) {
  SN_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2, "sn_chamfer_forward_host: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1 && m >= 1, "sn_chamfer_forward_host: need b,n,m >= 1 (got %d,%d,%d)", b, n, m);
  const long blocks1 = (n + kQueryBlock - 1) / kQueryBlock, blocks2 = (m + kQueryBlock - 1) / kQueryBlock;
  parallel_items((long)b * (blocks1 + blocks2), threads, [&](long item) {
    const long cloud = item / (blocks1 + blocks2), r = item % (blocks1 + blocks2);
    if (r < blocks1) {
      const int q0 = (int)r * kQueryBlock, q1 = std::min(n, q0 + kQueryBlock);
      nn_block(xyz1 + cloud * n * 3, q0, q1, xyz2 + cloud * m * 3, m, dist1 + cloud * n, idx1 + cloud * n);
    } else {
      const int q0 = (int)(r - blocks1) * kQueryBlock, q1 = std::min(m, q0 + kQueryBlock);
      nn_block(xyz2 + cloud * m * 3, q0, q1, xyz1 + cloud * n * 3, n, dist2 + cloud * m, idx2 + cloud * m);
    }
  });
  return 0;
}

extern "C" int sn_chamfer_backward_host(const float *xyz1, const float *xyz2, const float *graddist1,
                                        const float *graddist2, const int *idx1, const int *idx2, int b, int n,
                                        int m, float *gradxyz1, float *gradxyz2, int threads) {
  SN_REQUIRE(xyz1 && xyz2 && graddist1 && graddist2 && idx1 && idx2 && gradxyz1 && gradxyz2,
             "sn_chamfer_backward_host: null pointer");
  SN_REQUIRE(b >= 1 && n >= 1 && m >= 1, "sn_chamfer_backward_host: need b,n,m >= 1 (got %d,%d,%d)", b, n, m);
  for (long i = 0; i < (long)b * n; ++i)
    SN_REQUIRE((unsigned)idx1[i] < (unsigned)m, "sn_chamfer_backward_host: idx1[%ld] = %d outside [0, %d)", i, idx1[i], m);
  for (long i = 0; i < (long)b * m; ++i)
    SN_REQUIRE((unsigned)idx2[i] < (unsigned)n, "sn_chamfer_backward_host: idx2[%ld] = %d outside [0, %d)", i, idx2[i], n);
  parallel_items(b, threads, [&](long cloud) {
    const float *p1 = xyz1 + cloud * n * 3, *p2 = xyz2 + cloud * m * 3;
    float *g1 = gradxyz1 + cloud * n * 3, *g2 = gradxyz2 + cloud * m * 3;
    std::memset(g1, 0, sizeof(float) * 3 * n);
    std::memset(g2, 0, sizeof(float) * 3 * m);
    auto half = [](const float *a, const float *o, const float *gd, const int *ix, int cnt, float *ga, float *go) {
      for (int j = 0; j < cnt; ++j) {
        const int k = ix[j];
        const float g = gd[j] * 2;
        for (int c = 0; c < 3; ++c) {
          const float term = g * (a[j * 3 + c] - o[k * 3 + c]);
          ga[j * 3 + c] += term;
          go[k * 3 + c] -= term;
        }
      }
    };
    half(p1, p2, graddist1 + cloud * n, idx1 + cloud * n, n, g1, g2);   // chamfer_distance.cpp:143-160
    half(p2, p1, graddist2 + cloud * m, idx2 + cloud * m, m, g2, g1);   // :161-178
  });
  return 0;
}
