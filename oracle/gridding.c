/*
 * oracle/gridding.c -- TEST INFRASTRUCTURE (see sn_oracle.h).
 * CPU restatement of the GRNet grid ops the reference ships:
 *   gridding forward/backward          cuda/gridding/gridding.cu:29-177, :213-312
 *   gridding reverse forward/backward  cuda/gridding/gridding_reverse.cu:30-103, :124-214
 *   cubic feature sampling fwd/bwd     cuda/cubic_feature_sampling/cubic_feature_sampling.cu:29-102, :135-174
 * Points are visited in ascending index order, so the order-dependent fp32 atomic
 * sums of the GPU kernels become sequential sums here (tests compare those with a
 * tolerance; indices, weights and everything single-writer are compared exactly).
 * Corner order LLL, LLU, LUL, LUU, ULL, ULU, UUL, UUU (x major); lower = floor,
 * upper = ceil, upper += 1 when equal; per-axis weight 1 - |p - corner|.
 * Like the reference there is NO bounds check on the forward scatter index; the
 * restatement (and the HIP kernel) skip vertices outside [0, n_vertices) instead of
 * writing out of bounds.  Pinned by tests/golden/gridding_*.npz, cubic_*.npz
 * (reference kernel text under the SIMT emulator).
 */
#include "sn_oracle.h"
#include <math.h>
#include <stddef.h>

static void corners(float p, int *lo, int *up) {
  *lo = (int)floorf(p);
  *up = (int)ceilf(p);
  if (*lo == *up) *up += 1;
}

void oracle_gridding_forward(const float *ptcloud, int b, int npts, int scale, float *grid,
                             float *weights, int *indexes) {
  const int s = scale / 2, len = 2 * s;
  const int nverts = len * len * len;
  for (size_t e = 0; e < (size_t)b * nverts; ++e) grid[e] = 0.f;
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < npts; ++j) {
      const float *p = ptcloud + ((size_t)i * npts + j) * 3;
      int lo[3], up[3];
      for (int a = 0; a < 3; ++a) corners(p[a], &lo[a], &up[a]);
      float *w = weights + ((size_t)i * npts + j) * 24;
      int *ix = indexes + ((size_t)i * npts + j) * 8;
      for (int c = 0; c < 8; ++c) {
        const int cx = (c & 4) ? up[0] : lo[0], cy = (c & 2) ? up[1] : lo[1],
                  cz = (c & 1) ? up[2] : lo[2];
        ix[c] = (cx + s) * len * len + (cy + s) * len + (cz + s);
        w[c * 3 + 0] = 1 - fabsf(p[0] - cx);
        w[c * 3 + 1] = 1 - fabsf(p[1] - cy);
        w[c * 3 + 2] = 1 - fabsf(p[2] - cz);
      }
      for (int c = 0; c < 8; ++c)
        if (ix[c] >= 0 && ix[c] < nverts)
          grid[(size_t)i * nverts + ix[c]] += w[c * 3 + 0] * w[c * 3 + 1] * w[c * 3 + 2];
    }
}

/* cuda/gridding_loss/gridding_distance.cu:29-177 (forward kernel), :179-212 (launcher):
 * the gridding weights over an integer box, eight accumulators per vertex (slot =
 * vertex * 8 + corner).  Sequential accumulation in point order. */
void oracle_gridding_dist_forward(const float *ptcloud, int b, int npts, int min_x, int max_x,
                                  int min_y, int max_y, int min_z, int max_z, float *grid,
                                  float *weights, int *indexes) {
  const int len_y = max_y - min_y + 1, len_z = max_z - min_z + 1;
  const long nslots = (long)(max_x - min_x + 1) * len_y * len_z * 8;
  for (size_t e = 0; e < (size_t)b * nslots; ++e) grid[e] = 0.f;
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < npts; ++j) {
      const float *p = ptcloud + ((size_t)i * npts + j) * 3;
      int lo[3], up[3];
      for (int a = 0; a < 3; ++a) corners(p[a], &lo[a], &up[a]);
      float *w = weights + ((size_t)i * npts + j) * 24;
      int *ix = indexes + ((size_t)i * npts + j) * 8;
      for (int c = 0; c < 8; ++c) {
        const int cx = (c & 4) ? up[0] : lo[0], cy = (c & 2) ? up[1] : lo[1],
                  cz = (c & 1) ? up[2] : lo[2];
        ix[c] = (((cx - min_x) * len_y + (cy - min_y)) * len_z + (cz - min_z)) * 8 + c;
        w[c * 3 + 0] = 1 - fabsf(p[0] - cx);
        w[c * 3 + 1] = 1 - fabsf(p[1] - cy);
        w[c * 3 + 2] = 1 - fabsf(p[2] - cz);
      }
      for (int c = 0; c < 8; ++c)
        if (ix[c] >= 0 && ix[c] < nslots)
          grid[(size_t)i * nslots + ix[c]] += w[c * 3 + 0] * w[c * 3 + 1] * w[c * 3 + 2];
    }
}

void oracle_gridding_backward(const float *grad_grid, const float *weights, const int *indexes,
                              int b, int npts, int nverts, float *grad_ptcloud) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < npts; ++j) {
      const float *w = weights + ((size_t)i * npts + j) * 24;
      const int *ix = indexes + ((size_t)i * npts + j) * 8;
      float gx = 0.f, gy = 0.f, gz = 0.f;
      for (int c = 0; c < 8; ++c) {
        const float g = (ix[c] >= 0 && ix[c] < nverts) ? grad_grid[(size_t)i * nverts + ix[c]] : 0.f;
        const float wx = w[c * 3], wy = w[c * 3 + 1], wz = w[c * 3 + 2];
        const float sx = (c & 4) ? g : -g, sy = (c & 2) ? g : -g, sz = (c & 1) ? g : -g;
        gx += sx * wy * wz;
        gy += sy * wx * wz;
        gz += sz * wx * wy;
      }
      float *o = grad_ptcloud + ((size_t)i * npts + j) * 3;
      o[0] = gx;
      o[1] = gy;
      o[2] = gz;
    }
}

static int vidx(int x, int y, int z, int scale) { return x * scale * scale + y * scale + z; }

/* returns 0 when the vertex produces no point */
static int rev_setup(const float *g, int j, int scale, int idx[8], float w[8], float *wsum,
                     int off[3]) {
  const int sq = scale * scale;
  const int x = j / sq, y = j % sq / scale, z = j % sq % scale;
  if (x == 0 || y == 0 || z == 0) return 0;
  idx[0] = vidx(x - 1, y - 1, z - 1, scale);
  idx[1] = vidx(x - 1, y - 1, z, scale);
  idx[2] = vidx(x - 1, y, z - 1, scale);
  idx[3] = vidx(x - 1, y, z, scale);
  idx[4] = vidx(x, y - 1, z - 1, scale);
  idx[5] = vidx(x, y - 1, z, scale);
  idx[6] = vidx(x, y, z - 1, scale);
  idx[7] = j;
  float s = 0;
  for (int i = 0; i < 8; ++i) {
    w[i] = g[idx[i]];
    s += w[i];
  }
  if (s < 1e-6) return 0;
  for (int i = 0; i < 8; ++i) w[i] /= s;
  *wsum = s;
  off[0] = x - scale / 2;
  off[1] = y - scale / 2;
  off[2] = z - scale / 2;
  return 1;
}

void oracle_gridding_reverse_forward(const float *grid, int b, int scale, float *ptcloud) {
  const int n = scale * scale * scale;
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      float *o = ptcloud + ((size_t)i * n + j) * 3;
      o[0] = o[1] = o[2] = 0.f;
      int idx[8], off[3];
      float w[8], ws;
      if (!rev_setup(grid + (size_t)i * n, j, scale, idx, w, &ws, off)) continue;
      for (int a = 0; a < 3; ++a) {
        float acc = 0.f;
        for (int c = 0; c < 8; ++c) {
          const int hi = a == 0 ? (c & 4) : (a == 1 ? (c & 2) : (c & 1));
          const float coord = (float)(hi ? off[a] : off[a] - 1);
          acc = c == 0 ? w[c] * coord : acc + w[c] * coord;
        }
        o[a] = acc;
      }
    }
}

void oracle_gridding_reverse_backward(const float *grad_ptcloud, const float *grid,
                                      const float *ptcloud, int b, int scale, float *grad_grid) {
  const int n = scale * scale * scale;
  for (size_t e = 0; e < (size_t)b * n; ++e) grad_grid[e] = 0.f;
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      int idx[8], off[3];
      float w[8], ws;
      if (!rev_setup(grid + (size_t)i * n, j, scale, idx, w, &ws, off)) continue;
      const float *gp = grad_ptcloud + ((size_t)i * n + j) * 3;
      const float *pc = ptcloud + ((size_t)i * n + j) * 3;
      for (int c = 0; c < 8; ++c) {
        const float cx = (float)((c & 4) ? off[0] : off[0] - 1) - pc[0];
        const float cy = (float)((c & 2) ? off[1] : off[1] - 1) - pc[1];
        const float cz = (float)((c & 1) ? off[2] : off[2] - 1) - pc[2];
        grad_grid[(size_t)i * n + idx[c]] += gp[0] * cx / ws + gp[1] * cy / ws + gp[2] * cz / ws;
      }
    }
}

void oracle_cubic_forward(const float *ptcloud, const float *feat, int b, int npts, int c,
                          int scale, int ns, float *out, int *idx) {
  const int nv = (2 * ns) * (2 * ns) * (2 * ns), cub = scale * scale * scale;
  for (size_t e = 0; e < (size_t)b * npts * nv * c; ++e) out[e] = 0.f;
  for (int i = 0; i < b; ++i)
    for (int p = 0; p < npts; ++p) {
      const float *pt = ptcloud + ((size_t)i * npts + p) * 3;
      int lo[3], up[3];
      for (int a = 0; a < 3; ++a) corners(pt[a], &lo[a], &up[a]);
      int *ix = idx + ((size_t)i * npts + p) * nv;
      int v = 0;
      const int e = ns - 1;
      for (int j = lo[0] - e; j <= up[0] + e; ++j)
        for (int k = lo[1] - e; k <= up[1] + e; ++k)
          for (int m = lo[2] - e; m <= up[2] + e; ++m)
            ix[v++] = (j < 0 || j >= scale || k < 0 || k >= scale || m < 0 || m >= scale)
                          ? -1
                          : vidx(j, k, m, scale);
      for (int j = 0; j < nv; ++j) {
        if (ix[j] == -1) continue;
        for (int k = 0; k < c; ++k)
          out[(((size_t)i * npts + p) * nv + j) * c + k] = feat[((size_t)i * c + k) * cub + ix[j]];
      }
    }
}

void oracle_cubic_backward(const float *grad_out, const int *idx, int b, int npts, int c,
                           int scale, int ns, float *grad_feat) {
  const int nv = (2 * ns) * (2 * ns) * (2 * ns), cub = scale * scale * scale;
  for (size_t e = 0; e < (size_t)b * c * cub; ++e) grad_feat[e] = 0.f;
  for (int i = 0; i < b; ++i)
    for (int p = 0; p < npts; ++p)
      for (int j = 0; j < nv; ++j) {
        const int v = idx[((size_t)i * npts + p) * nv + j];
        if (v == -1) continue;
        for (int k = 0; k < c; ++k)
          grad_feat[((size_t)i * c + k) * cub + v] +=
              grad_out[(((size_t)i * npts + p) * nv + j) * c + k];
      }
}
