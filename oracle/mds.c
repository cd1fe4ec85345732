/*
 * oracle/mds.c -- TEST INFRASTRUCTURE (see sn_oracle.h).
 * CPU restatement of minimum density sampling and gather
 * (cuda/MDS/MDS_cuda.cu:91-211 sampling kernel, :29-41 / :55-69 gather fwd/bwd),
 * with the INTENDED race-free semantics (the kernel has two formal races,
 * :200 vs :134-135 and :203 vs :130, that are benign on lock-step hardware):
 *   bs = min(2^floor(log2 n), 1024)                                  (:8-12)
 *   t  = (float)(5.0 * (double)mml * (double)mml)                    (:114)
 *   idx[0] = 0, temp[0] = 1e9
 *   every round: temp[k] = (float)((double)temp[k] + w), w = e (k < 8192) or
 *   (double)e * 2.0, e = exp(-d/t) as FLOAT, d = (dx*dx+dy*dy)+dz*dz  (:128-130)
 *   pick = argmin temp; ties: per thread lowest k (strict '<', :131-132), across
 *   threads the tree of :139-198 keeps the LOWER slot on ties, which orders
 *   threads by bit-reversed tid  =>  argmin (temp, bitrev(k mod bs), k).
 * exp_mode 0: libm expf (the textual semantics); exp_mode 1: sn_expf
 * (include/sn_expf.h), the function the HIP kernel uses -- see that header for
 * why bit parity needs a shared exponential.
 * Pinned: accumulate / threshold / pick / 1e9 logic against the reference kernel's
 * single-thread instantiation run by the emulator (tests/golden/mds_*.npz, bs=1);
 * the cross-thread tie order is a pure function of the reduction tree and is
 * tested by simulating that tree (tests/test_mds.py).
 */
#include "sn_oracle.h"
#include "../include/sn_expf.h"
#include <math.h>
#include <stdlib.h>

static int opt_threads(int n) {
  int p = 1;
  while (p * 2 <= n && p < 1024) p *= 2;
  return p;
}

static unsigned bitrev(unsigned v, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

void oracle_mds(const float *xyz, int b, int n, int m, const float *mean_mst_length,
                int exp_mode, int bs_override, int *idx) {
  if (m <= 0) return;
  const int bs = bs_override > 0 ? bs_override : opt_threads(n);
  int lg = 0;
  while ((1 << lg) < bs) ++lg;
  unsigned *rank = (unsigned *)malloc(sizeof(unsigned) * bs);
  for (int t = 0; t < bs; ++t) rank[t] = bitrev((unsigned)t, lg);
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < b; ++i) {
    const float *p = xyz + (size_t)i * n * 3;
    int *out = idx + (size_t)i * m;
    float *temp = (float *)calloc(n, sizeof(float));
    const float t = (float)(5.0 * (double)mean_mst_length[i] * (double)mean_mst_length[i]);
    int old = 0;
    out[0] = 0;
    temp[0] = 1e9f;
    for (int j = 1; j < m; ++j) {
      const float x1 = p[old * 3], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
      float best = 1e9f;
      unsigned best_rank = 0;
      int besti = 0, have = 0;
      for (int k = 0; k < n; ++k) {
        const float dx = p[k * 3] - x1, dy = p[k * 3 + 1] - y1, dz = p[k * 3 + 2] - z1;
        const float d = dx * dx + dy * dy + dz * dz;
        const float a = -d / t;
        const float e = exp_mode ? sn_expf(a) : expf(a);
        const double w = k < 8192 ? (double)e : (double)e * 2.0;
        temp[k] = (float)((double)temp[k] + w);
        /* argmin (temp, bitrev(tid), k); per-thread candidates start at (1e9, 0) */
        const float v = temp[k];
        if (v < 1e9f) {
          const unsigned rk = rank[k % bs];
          if (!have || v < best || (v == best && rk < best_rank)) {
            best = v;
            best_rank = rk;
            besti = k;
            have = 1;
          }
        }
      }
      old = have ? besti : 0; /* all >= 1e9: every thread reports (1e9, 0) -> index 0 */
      out[j] = old;
      temp[old] = 1e9f;
    }
    free(temp);
  }
  free(rank);
}

void oracle_gather_forward(const float *feat, const int *idx, int b, int c, int n, int m,
                           float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        out[((size_t)i * c + l) * m + j] = feat[((size_t)i * c + l) * n + idx[(size_t)i * m + j]];
}

void oracle_gather_backward(const float *grad_out, const int *idx, int b, int c, int n, int m,
                            float *grad_feat) {
  for (size_t e = 0; e < (size_t)b * c * n; ++e) grad_feat[e] = 0.f;
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        grad_feat[((size_t)i * c + l) * n + idx[(size_t)i * m + j]] +=
            grad_out[((size_t)i * c + l) * m + j];
}
