/*
 * oracle/expansion.c -- TEST INFRASTRUCTURE (see sn_oracle.h).
 * CPU restatement of the reference expansion penalty
 * (cuda/expansion_penalty/expansion_penalty_cuda.cu:7-149 forward, :167-184
 * backward).  Per patch of P consecutive points (P a power of two, <= 512):
 *   1. Prim's MST from vertex 0 on Euclidean (sqrtf) lengths; the next vertex is
 *      the arg-min of cur_dis through the reference's reduction tree, in which
 *      the RIGHT slot survives ties (:64-73)  => highest index among ties;
 *      cur_dis updates use strict '<' (:53) => earliest parent kept on ties.
 *   2. mean_dis = tree_sum(edge lengths) / (P-1) in the pairwise up-sweep order
 *      of :103-110; mean_mst_length[b] accumulates mean_dis over patches in
 *      ascending patch order (the reference uses an fp32 atomicAdd: order free
 *      on the GPU; ascending is what a sequential run gives) and is returned
 *      UN-normalised, exactly what the reference kernel leaves in the tensor --
 *      the Python module divides by n/P afterwards (module :40).
 *   3. leaf stripping (:120-147) with SNAPSHOT semantics: all cnt[] reads of a
 *      round precede its decrements.  (On the GPU the reads are live and the
 *      owner of the last star's final edge is timing dependent; see DESIGN.md.)
 *      Observed: the reference kernel text run under tests/golden/gen/simt.h with randomised thread
 *      schedules changes dist / assignment on tests/golden/xfail_expansion_rand_3x64_P16.npz (one edge's
 *      owner flips between its two endpoints); the multiset of penalised lengths and the mean MST
 *      length do not change -- those are what the loss uses.
 */
#include "sn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PMAX 512

static void patch_forward(const float *x, int P, float alpha, int base,
                          float *dist, int *assignment, float *mean_out) {
  int vis[PMAX], cur_idx[PMAX], parent[PMAX], cnt[PMAX];
  float cur_dis[PMAX], w[PMAX], red[PMAX];
  int red_i[PMAX];
  for (int v = 0; v < P; ++v) {
    vis[v] = 0;
    cur_dis[v] = 1e9f;
    cur_idx[v] = 0;
    cnt[v] = 0;
    parent[v] = -1;
    w[v] = 0.f;
  }
  vis[0] = 1;
  int last = 0;
  for (int round = 0; round < P - 1; ++round) {
    const float xl = x[last * 3], yl = x[last * 3 + 1], zl = x[last * 3 + 2];
    for (int v = 0; v < P; ++v) {
      if (!vis[v]) {
        const float dx = x[v * 3] - xl, dy = x[v * 3 + 1] - yl, dz = x[v * 3 + 2] - zl;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if (d < cur_dis[v]) {
          cur_dis[v] = d;
          cur_idx[v] = last;
        }
        red[v] = cur_dis[v];
      } else {
        red[v] = 1e9f;
      }
      red_i[v] = v;
    }
    for (int stride = 1; stride <= P / 2; stride *= 2)
      for (int t = 0; t < P; ++t) {
        const int index = (t + 1) * stride * 2 - 1;
        if (index < P && red[index - stride] < red[index]) {
          red[index] = red[index - stride];
          red_i[index] = red_i[index - stride];
        }
      }
    last = red_i[P - 1];
    vis[last] = 1;
    parent[last] = cur_idx[last];
    w[last] = cur_dis[last];
    cnt[last] += 1;
    cnt[parent[last]] += 1;
  }
  /* mean edge length, pairwise up-sweep */
  for (int v = 0; v < P; ++v) red[v] = w[v];
  for (int stride = 1; stride <= P / 2; stride *= 2)
    for (int t = 0; t < P; ++t) {
      const int index = (t + 1) * stride * 2 - 1;
      if (index < P) red[index] += red[index - stride];
    }
  const float mean_dis = red[P - 1] / (P - 1);
  *mean_out = mean_dis;
  for (int v = 0; v < P; ++v) {
    dist[v] = 0.f;
    assignment[v] = -1;
  }
  /* leaf stripping over the P-1 tree edges (child c, parent[c]) */
  int alive[PMAX], snap[PMAX];
  for (int v = 0; v < P; ++v) alive[v] = (parent[v] >= 0);
  const float thr = mean_dis * alpha;
  for (;;) {
    int leaves = 0;
    memcpy(snap, cnt, sizeof(int) * P);
    for (int v = 0; v < P; ++v) leaves += (snap[v] == 1);
    if (!leaves) break;
    for (int c = 0; c < P; ++c) {
      if (!alive[c]) continue;
      const int p = parent[c];
      int owner = -1, other = -1;
      if (snap[c] == 1 && (snap[p] > 1 || (snap[p] == 1 && c > p))) {
        owner = c;
        other = p;
      } else if (snap[p] == 1 && (snap[c] > 1 || (snap[c] == 1 && p > c))) {
        owner = p;
        other = c;
      }
      if (owner >= 0) {
        alive[c] = 0;
        cnt[c] -= 1;
        cnt[p] -= 1;
        if (w[c] > thr) {
          dist[owner] = w[c];
          assignment[owner] = base + other;
        }
      }
    }
  }
}

void oracle_expansion_forward(const float *xyz, int b, int n, int primitive_size,
                              float alpha, float *dist, int *assignment,
                              float *mean_mst_length) {
  const int P = primitive_size, np = n / P;
  for (int i = 0; i < b; ++i) {
    float acc = 0.f;
    for (int p = 0; p < np; ++p) {
      float m;
      const size_t off = (size_t)i * n + (size_t)p * P;
      patch_forward(xyz + off * 3, P, alpha, p * P, dist + off, assignment + off, &m);
      acc += m;
    }
    mean_mst_length[i] = acc;
  }
}

void oracle_expansion_backward(const float *xyz, const float *graddist,
                               const int *assignment, int b, int n, float *gradxyz) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const size_t e = (size_t)i * n + j;
      float *g = gradxyz + e * 3;
      g[0] = g[1] = g[2] = 0.f;
      const int j2 = assignment[e];
      if (j2 == -1) continue;
      const float *a = xyz + e * 3, *o = xyz + ((size_t)i * n + j2) * 3;
      const float gg = graddist[e] * 2;
      g[0] = gg * (a[0] - o[0]);
      g[1] = gg * (a[1] - o[1]);
      g[2] = gg * (a[2] - o[2]);
    }
}
