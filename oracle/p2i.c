/*
 * oracle/p2i.c -- TEST INFRASTRUCTURE (see sn_oracle.h).
 * CPU restatement of the reference point-to-image splat ("p2i"):
 *   max forward/backward  cuda/p2i_op/p2i_max.h:7-66, :68-143
 *   sum forward/backward  cuda/p2i_op/p2i_sum.h:7-58, :60-131
 *   pixel walk            cuda/p2i_op/utility.h:82-100 (x outer, y inner,
 *                         clamp(floor(p-R)) .. clamp(ceil(p+R)), r <= R)
 * Points arrive in pixel space (row, col), as cuda/p2i_op/__init__.py:117-121
 * hands them to the op.  The cosine weight is evaluated in DOUBLE even for
 * float tensors (r * M_PI / R with M_PI a double literal) and then narrowed.
 * Order: the reference's sequential cpu_device launcher (common.h:55-78) runs
 * ids ascending, so equal splat values keep the LOWEST point id (strict '<');
 * on the GPU that order is a race -- lowest id is the canonical rule here.
 * Pinned by tests/golden/p2i_*.npz = outputs of the reference functors compiled
 * for the CPU (tests/golden/gen_p2i.py) plus the 8x8 known answer of
 * cuda/p2i_op/p2i_test.py:10-20.
 */
#include "sn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct {
  int min_x, max_x, min_y, max_y;
} box_t;

static box_t box_of(float py, float px, int h, int w, float radius) {
  box_t b;
  b.min_x = clampi((int)floorf(px - radius), 0, w - 1);
  b.max_x = clampi((int)ceilf(px + radius), 0, w - 1);
  b.min_y = clampi((int)floorf(py - radius), 0, h - 1);
  b.max_y = clampi((int)ceilf(py + radius), 0, h - 1);
  return b;
}

static float cos_weight(float r, float radius) {
  return (float)(cos((double)r * M_PI / (double)radius) * 0.5 + 0.5);
}

void oracle_p2i_max_forward(const float *points, const float *feat, const int *batch_inds,
                            int npoints, int channels, int batch, int h, int w, float radius,
                            float *out, int *out_ids) {
  for (int id = 0; id < npoints * channels; ++id) {
    const int c = id % channels, pid = (id / channels) % npoints;
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const box_t bx = box_of(py, px, h, w, radius);
    for (int x = bx.min_x; x <= bx.max_x; ++x)
      for (int y = bx.min_y; y <= bx.max_y; ++y) {
        const float dx = x - px, dy = y - py;
        const float r = sqrtf(dx * dx + dy * dy);
        if (!(r <= radius)) continue;
        const size_t index = (((size_t)b * channels + c) * h + y) * w + x;
        const float v = feat[id] * cos_weight(r, radius);
        if (out[index] < v) {
          out[index] = v;
          out_ids[index] = pid;
        }
      }
  }
}

/* The same splat with the images of different clouds painted by different threads: points are visited
 * in ascending id inside every cloud, so every pixel sees exactly the sequence of candidates it sees in
 * oracle_p2i_max_forward (bit-identical output; bench.py's multi-threaded CPU baseline). */
void oracle_p2i_max_forward_mt(const float *points, const float *feat, const int *batch_inds,
                               int npoints, int channels, int batch, int h, int w, float radius,
                               float *out, int *out_ids) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int bb = 0; bb < batch; ++bb)
    for (int id = 0; id < npoints * channels; ++id) {
      const int c = id % channels, pid = (id / channels) % npoints;
      if (batch_inds[pid] != bb) continue;
      const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
      const box_t bx = box_of(py, px, h, w, radius);
      for (int x = bx.min_x; x <= bx.max_x; ++x)
        for (int y = bx.min_y; y <= bx.max_y; ++y) {
          const float dx = x - px, dy = y - py;
          const float r = sqrtf(dx * dx + dy * dy);
          if (!(r <= radius)) continue;
          const size_t index = (((size_t)bb * channels + c) * h + y) * w + x;
          const float v = feat[id] * cos_weight(r, radius);
          if (out[index] < v) {
            out[index] = v;
            out_ids[index] = pid;
          }
        }
    }
}

void oracle_p2i_max_backward(const float *out_grad, const int *out_ids, const float *points,
                             const float *feat, int npoints, int channels, int batch, int h,
                             int w, float radius, float *points_grad, float *feat_grad,
                             float *background_grad) {
  for (int i = 0; i < npoints * 2; ++i) points_grad[i] = 0.f;
  for (int i = 0; i < npoints * channels; ++i) feat_grad[i] = 0.f;
  const size_t total = (size_t)batch * channels * h * w;
  for (size_t index = 0; index < total; ++index) {
    background_grad[index] = 0.f;
    const int x = (int)(index % w), y = (int)((index / w) % h);
    const int c = (int)((index / ((size_t)w * h)) % channels);
    const float g = out_grad[index];
    const int pid = out_ids[index];
    if (pid < 0) {
      background_grad[index] += g;
      continue;
    }
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const float dx = x - px, dy = y - py;
    const float r = sqrtf(dx * dx + dy * dy);
    const float wgt = cos_weight(r, radius);
    const float fv = feat[pid * channels + c];
    feat_grad[pid * channels + c] += g * wgt;
    const float wg = g * fv;
    const float rm = r > 1e-10f ? r : 1e-10f;
    const float k = (float)((double)wg * sin((double)r * M_PI / (double)radius) * 0.5 * M_PI /
                            (double)radius / (double)rm);
    points_grad[pid * 2 + 0] += k * dy;
    points_grad[pid * 2 + 1] += k * dx;
  }
}

// The same fp32 TERMS as oracle_p2i_max_backward (p2i_max.h:94-143), accumulated without rounding: every term is a
// float, the sums run in double (exact for < 2^29 terms of comparable size) and are rounded to float once.
// Test infrastructure: what the reference's sequential fp32 `+=` (above) and any order of fp32 atomics on a GPU
// both approximate, each within (#terms) 2^-24 sum|terms| -- the tolerance the tests give the fp32 order -- and what
// the HIP path's 64-bit fixed-point accumulation reproduces to the last bit or two.
void oracle_p2i_max_backward_exact(const float *out_grad, const int *out_ids, const float *points,
                                   const float *feat, int npoints, int channels, int batch, int h,
                                   int w, float radius, float *points_grad, float *feat_grad) {
  double *pg = (double *)calloc((size_t)npoints * 2, sizeof(double));
  double *fg = (double *)calloc((size_t)npoints * channels, sizeof(double));
  const size_t total = (size_t)batch * channels * h * w;
  for (size_t index = 0; index < total; ++index) {
    const int x = (int)(index % w), y = (int)((index / w) % h);
    const int c = (int)((index / ((size_t)w * h)) % channels);
    const float g = out_grad[index];
    const int pid = out_ids[index];
    if (pid < 0) continue;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const float dx = x - px, dy = y - py;
    const float r = sqrtf(dx * dx + dy * dy);
    const float wgt = cos_weight(r, radius);
    const float fv = feat[pid * channels + c];
    fg[pid * channels + c] += (double)(float)(g * wgt);
    const float wg = g * fv;
    const float rm = r > 1e-10f ? r : 1e-10f;
    const float k = (float)((double)wg * sin((double)r * M_PI / (double)radius) * 0.5 * M_PI /
                            (double)radius / (double)rm);
    pg[pid * 2 + 0] += (double)(float)(k * dy);
    pg[pid * 2 + 1] += (double)(float)(k * dx);
  }
  for (int i = 0; i < npoints * 2; ++i) points_grad[i] = (float)pg[i];
  for (int i = 0; i < npoints * channels; ++i) feat_grad[i] = (float)fg[i];
  free(pg);
  free(fg);
}

void oracle_p2i_sum_forward(const float *points, const float *feat, const int *batch_inds,
                            int npoints, int channels, int batch, int h, int w, float radius,
                            float *out) {
  for (int id = 0; id < npoints * channels; ++id) {
    const int c = id % channels, pid = (id / channels) % npoints;
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const box_t bx = box_of(py, px, h, w, radius);
    for (int x = bx.min_x; x <= bx.max_x; ++x)
      for (int y = bx.min_y; y <= bx.max_y; ++y) {
        const float dx = x - px, dy = y - py;
        const float r = sqrtf(dx * dx + dy * dy);
        if (!(r <= radius)) continue;
        const size_t index = (((size_t)b * channels + c) * h + y) * w + x;
        out[index] += cos_weight(r, radius) * feat[id];
      }
  }
}

void oracle_p2i_sum_backward(const float *out_grad, const float *points, const float *feat,
                             const int *batch_inds, int npoints, int channels, int batch, int h,
                             int w, float radius, float *points_grad, float *feat_grad) {
  for (int i = 0; i < npoints * 2; ++i) points_grad[i] = 0.f;
  for (int i = 0; i < npoints * channels; ++i) feat_grad[i] = 0.f;
  for (int id = 0; id < npoints * channels; ++id) {
    const int c = id % channels, pid = (id / channels) % npoints;
    const int b = batch_inds[pid];
    if (b < 0 || b >= batch) continue;
    const float py = points[pid * 2 + 0], px = points[pid * 2 + 1];
    const box_t bx = box_of(py, px, h, w, radius);
    for (int x = bx.min_x; x <= bx.max_x; ++x)
      for (int y = bx.min_y; y <= bx.max_y; ++y) {
        const float dx = x - px, dy = y - py;
        const float r = sqrtf(dx * dx + dy * dy);
        if (!(r <= radius)) continue;
        const float wgt = cos_weight(r, radius);
        const float fv = feat[id];
        const size_t index = (((size_t)b * channels + c) * h + y) * w + x;
        const float g = out_grad[index];
        feat_grad[id] += g * wgt;
        const float wg = g * fv;
        const float rm = r > 1e-10f ? r : 1e-10f;
        const double s = (double)wg * sin((double)r * M_PI / (double)radius) * 0.5 * M_PI /
                         (double)radius;
        points_grad[pid * 2 + 0] += (float)(s * (double)dy / (double)rm);
        points_grad[pid * 2 + 1] += (float)(s * (double)dx / (double)rm);
      }
  }
}
