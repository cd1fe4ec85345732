/*
 * oracle/emd.c -- TEST INFRASTRUCTURE (see sn_oracle.h).
 * Sequential CPU restatement of the reference's auction-based EMD
 * (cuda/emd/emd_cuda.cu:23-282, scratch initialisation cuda/emd/emd_module.py:41-54).
 *
 * The reference is a GPU program with two order-dependent spots; this file fixes
 * the order the way a sequential execution of the same code (threads in
 * ascending index order) resolves them, and the HIP kernels implement the same
 * rules deterministically:
 *   Bid (emd_cuda.cu:95-179): a bidder's thread group splits every 2048-tile in
 *     contiguous chunks (:136-139); each thread keeps (best, better, best_i)
 *     with strict '>' and thread 0 merges the group in ascending thread order
 *     with strict '>' (:166-173).  Restated literally below, so exact ties in
 *     the bid value resolve to argmin (thread(k), k) exactly as in the kernel.
 *   GetMax (:181-194): every bidder within +-1e-6 (double compare) of the
 *     target's maximum increment writes max_idx; "last writer wins" is a race on
 *     the GPU, here: ascending j, i.e. the HIGHEST bidder index in the window.
 * Arithmetic: bid value d = (float)((3.0 - (double)sqrtf(s)) - (double)price),
 * s = (dx*dx + dy*dy) + dz*dz rounded per operation (build: -ffp-contract=off).
 *
 * Pinning: tests/golden/emd_*.npz hold outputs of the reference's CUDA kernel
 * text executed by a SIMT-on-CPU emulator (tests/golden/gen_emulated.py); see
 * oracle/README.md for what that does and does not pin.
 */
#include "sn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float best, better;
  int best_i;
} top2_t;

/* one bidder, literal thread-group structure of Bid */
static void bid_one(const float *xyz2, const float *price, int n, float x1,
                    float y1, float z1, int tpu, top2_t *out) {
  const int batch = 2048;
  float best = -1e9f, better = -1e9f;
  int best_i = -1;
  /* thread 0 first (its own running values seed the merge), then 1..tpu-1 */
  for (int t = 0; t < tpu; ++t) {
    float tb = -1e9f, tbb = -1e9f;
    int ti = -1;
    for (int k2 = 0; k2 < n; k2 += batch) {
      const int end_k = (n < k2 + batch ? n : k2 + batch) - k2;
      const int delta = (end_k + tpu - 1) / tpu;
      const int l = t * delta;
      int r = (t + 1) * delta;
      if (r > end_k) r = end_k;
      for (int k = l; k < r; ++k) {
        const float *p = xyz2 + (size_t)(k2 + k) * 3;
        const float x2 = p[0] - x1, y2 = p[1] - y1, z2 = p[2] - z1;
        const float s = x2 * x2 + y2 * y2 + z2 * z2;
        const float d = (float)((3.0 - (double)sqrtf(s)) - (double)price[k2 + k]);
        if (d > tb) {
          tbb = tb;
          tb = d;
          ti = k + k2;
        } else if (d > tbb) {
          tbb = d;
        }
      }
    }
    if (t == 0) {
      best = tb;
      better = tbb;
      best_i = ti;
    } else if (tb > best) {
      better = best > tbb ? best : tbb;
      best = tb;
      best_i = ti;
    } else {
      better = better > tb ? better : tb;
    }
  }
  out->best = best;
  out->better = better;
  out->best_i = best_i;
}

static long long emd_impl(const float *xyz1, const float *xyz2, int b, int n,
                          float eps, int iters, float *dist, int *assignment,
                          float *price_out, int *trace_unass, int mt) {
  long long pairs = 0;
  if (trace_unass)
    for (int it = 0; it < iters; ++it) trace_unass[it] = 0;
  const int block_cnt = n / 1024;
  for (int i = 0; i < b; ++i) {
    const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * n * 3;
    int *assign = assignment + (size_t)i * n;
    int *assign_inv = (int *)malloc(sizeof(int) * n);
    float *price = (float *)calloc(n, sizeof(float));
    int *bid = (int *)calloc(n, sizeof(int));
    float *bid_inc = (float *)calloc(n, sizeof(float));
    float *max_inc = (float *)calloc(n, sizeof(float)); /* starts at 0, emd_module.py:49 */
    int *max_idx = (int *)calloc(n, sizeof(int));
    int *unass = (int *)malloc(sizeof(int) * n);
    for (int j = 0; j < n; ++j) assign[j] = assign_inv[j] = -1;

    for (int it = 0; it < iters; ++it) {
      int cnt = 0;
      for (int j = 0; j < n; ++j)
        if (assign[j] == -1) unass[cnt++] = j;
      if (trace_unass) trace_unass[it] += cnt;
      if (cnt == 0) continue; /* emd_cuda.cu:105-106 and all later kernels no-op */
      pairs += (long long)cnt * n;
      const int unass_per_block = (cnt + block_cnt - 1) / block_cnt;
      const int tpu = 1024 / unass_per_block;
      /* ---- Bid */
      if (mt) {
#pragma omp parallel for schedule(dynamic, 8)
        for (int u = 0; u < cnt; ++u) {
          const int j = unass[u];
          top2_t r;
          bid_one(p2, price, n, p1[j * 3], p1[j * 3 + 1], p1[j * 3 + 2], tpu, &r);
          bid[j] = r.best_i;
          bid_inc[j] = r.best - r.better + eps;
        }
      } else {
        for (int u = 0; u < cnt; ++u) {
          const int j = unass[u];
          top2_t r;
          bid_one(p2, price, n, p1[j * 3], p1[j * 3 + 1], p1[j * 3 + 2], tpu, &r);
          bid[j] = r.best_i;
          bid_inc[j] = r.best - r.better + eps;
        }
      }
      for (int u = 0; u < cnt; ++u) { /* atomicMax, order free */
        const int j = unass[u];
        if (bid_inc[j] > max_inc[bid[j]]) max_inc[bid[j]] = bid_inc[j];
      }
      /* ---- GetMax: ascending j, last writer inside the window wins */
      for (int u = 0; u < cnt; ++u) {
        const int j = unass[u];
        const float bi = bid_inc[j], mi = max_inc[bid[j]];
        if (bi - 1e-6 <= mi && mi <= bi + 1e-6) max_idx[bid[j]] = j;
      }
      /* ---- Assign */
      const int last = (it == iters - 1);
      for (int u = 0; u < cnt; ++u) {
        const int j = unass[u];
        const int t = bid[j];
        if (last || max_idx[t] == j) {
          const int inv = assign_inv[t];
          if (!last && inv != -1) assign[inv] = -1;
          assign_inv[t] = j;
          assign[j] = t;
          price[t] += bid_inc[j];
          max_inc[t] = -1e9f;
        }
      }
    }
    /* ---- CalcDist (emd_cuda.cu:217-226): xyz1 minus xyz2 */
    for (int j = 0; j < n; ++j) {
      const int k = assign[j];
      if (k < 0) { /* iters == 0: the reference reads out of bounds; define 0 */
        dist[(size_t)i * n + j] = 0.f;
        continue;
      }
      const float dx = p1[j * 3] - p2[k * 3], dy = p1[j * 3 + 1] - p2[k * 3 + 1],
                  dz = p1[j * 3 + 2] - p2[k * 3 + 2];
      dist[(size_t)i * n + j] = dx * dx + dy * dy + dz * dz;
    }
    if (price_out) memcpy(price_out + (size_t)i * n, price, sizeof(float) * n);
    free(assign_inv);
    free(price);
    free(bid);
    free(bid_inc);
    free(max_inc);
    free(max_idx);
    free(unass);
  }
  return pairs;
}

long long oracle_emd_forward(const float *xyz1, const float *xyz2, int b, int n,
                             float eps, int iters, float *dist, int *assignment,
                             float *price_out, int *trace_unass) {
  return emd_impl(xyz1, xyz2, b, n, eps, iters, dist, assignment, price_out, trace_unass, 0);
}

long long oracle_emd_forward_mt(const float *xyz1, const float *xyz2, int b, int n,
                                float eps, int iters, float *dist, int *assignment,
                                float *price_out, int *trace_unass) {
  return emd_impl(xyz1, xyz2, b, n, eps, iters, dist, assignment, price_out, trace_unass, 1);
}

void oracle_emd_backward(const float *xyz1, const float *xyz2, const float *graddist,
                         const int *assignment, int b, int n, float *gradxyz1) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const size_t e = (size_t)i * n + j;
      const float *a = xyz1 + e * 3;
      const float *o = xyz2 + ((size_t)i * n + assignment[e]) * 3;
      const float g = graddist[e] * 2;
      gradxyz1[e * 3 + 0] = g * (a[0] - o[0]);
      gradxyz1[e * 3 + 1] = g * (a[1] - o[1]);
      gradxyz1[e * 3 + 2] = g * (a[2] - o[2]);
    }
}
