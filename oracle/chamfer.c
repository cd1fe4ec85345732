/*
 * oracle/chamfer.c -- TEST INFRASTRUCTURE (see sn_oracle.h).
 * CPU restatement of the reference Chamfer distance.
 *   forward : cuda/chamfer_distance/chamfer_distance.cpp:57-112 (nnsearch),
 *             same semantics as the CUDA kernel chamfer_distance.cu:7-137
 *             (strict '<' inside a tile, strict '>' across tiles => lowest k
 *             attaining the minimum).
 *   backward: cuda/chamfer_distance/chamfer_distance.cpp:114-180.
 * Build with -ffp-contract=off: d = (dx*dx + dy*dy) + dz*dz with three
 * separately rounded products, as the reference's g++ build evaluates it.
 * Pinned against oracle/_ref (the reference's own CPU path compiled from
 * /root/reference) and tests/golden/chamfer_*.npz.
 */
#include "sn_oracle.h"
#include <stddef.h>

static void nn_one_dir(const float *q, const float *t, int n, int m,
                       float *dist, int *idx, int j0, int j1) {
  for (int j = j0; j < j1; ++j) {
    const float x1 = q[j * 3 + 0], y1 = q[j * 3 + 1], z1 = q[j * 3 + 2];
    float best = 0.0f;
    int besti = 0;
    for (int k = 0; k < m; ++k) {
      const float dx = t[k * 3 + 0] - x1; /* target minus query */
      const float dy = t[k * 3 + 1] - y1;
      const float dz = t[k * 3 + 2] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      if (k == 0 || d < best) {
        best = d;
        besti = k;
      }
    }
    dist[j] = best;
    idx[j] = besti;
  }
  (void)n;
}

void oracle_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n,
                            int m, float *dist1, int *idx1, float *dist2,
                            int *idx2) {
  for (int i = 0; i < b; ++i) {
    nn_one_dir(xyz1 + (size_t)i * n * 3, xyz2 + (size_t)i * m * 3, n, m,
               dist1 + (size_t)i * n, idx1 + (size_t)i * n, 0, n);
    nn_one_dir(xyz2 + (size_t)i * m * 3, xyz1 + (size_t)i * n * 3, m, n,
               dist2 + (size_t)i * m, idx2 + (size_t)i * m, 0, m);
  }
}

void oracle_chamfer_forward_mt(const float *xyz1, const float *xyz2, int b,
                               int n, int m, float *dist1, int *idx1,
                               float *dist2, int *idx2) {
  const int blk = 256;
  const int nb1 = (n + blk - 1) / blk, nb2 = (m + blk - 1) / blk;
  const long total = (long)b * (nb1 + nb2);
#pragma omp parallel for schedule(dynamic, 4)
  for (long w = 0; w < total; ++w) {
    const int i = (int)(w / (nb1 + nb2));
    const int r = (int)(w % (nb1 + nb2));
    if (r < nb1) {
      const int j0 = r * blk, j1 = (j0 + blk < n) ? j0 + blk : n;
      nn_one_dir(xyz1 + (size_t)i * n * 3, xyz2 + (size_t)i * m * 3, n, m,
                 dist1 + (size_t)i * n, idx1 + (size_t)i * n, j0, j1);
    } else {
      const int j0 = (r - nb1) * blk, j1 = (j0 + blk < m) ? j0 + blk : m;
      nn_one_dir(xyz2 + (size_t)i * m * 3, xyz1 + (size_t)i * n * 3, m, n,
                 dist2 + (size_t)i * m, idx2 + (size_t)i * m, j0, j1);
    }
  }
}

static void grad_one_dir(const float *a, const float *bq, const float *gd,
                         const int *idx, int n, float *ga, float *gb) {
  for (int j = 0; j < n; ++j) {
    const float x1 = a[j * 3 + 0], y1 = a[j * 3 + 1], z1 = a[j * 3 + 2];
    const int j2 = idx[j];
    const float x2 = bq[j2 * 3 + 0], y2 = bq[j2 * 3 + 1], z2 = bq[j2 * 3 + 2];
    const float g = gd[j] * 2;
    ga[j * 3 + 0] += g * (x1 - x2);
    ga[j * 3 + 1] += g * (y1 - y2);
    ga[j * 3 + 2] += g * (z1 - z2);
    gb[j2 * 3 + 0] -= (g * (x1 - x2));
    gb[j2 * 3 + 1] -= (g * (y1 - y2));
    gb[j2 * 3 + 2] -= (g * (z1 - z2));
  }
}

void oracle_chamfer_backward(const float *xyz1, const float *xyz2,
                             const float *graddist1, const float *graddist2,
                             const int *idx1, const int *idx2, int b, int n,
                             int m, float *gradxyz1, float *gradxyz2) {
  for (size_t i = 0; i < (size_t)b * n * 3; ++i) gradxyz1[i] = 0;
  for (size_t i = 0; i < (size_t)b * m * 3; ++i) gradxyz2[i] = 0;
  for (int i = 0; i < b; ++i) {
    const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
    float *g1 = gradxyz1 + (size_t)i * n * 3, *g2 = gradxyz2 + (size_t)i * m * 3;
    grad_one_dir(p1, p2, graddist1 + (size_t)i * n, idx1 + (size_t)i * n, n, g1, g2);
    grad_one_dir(p2, p1, graddist2 + (size_t)i * m, idx2 + (size_t)i * m, m, g2, g1);
  }
}
