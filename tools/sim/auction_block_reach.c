// blocks of 16 targets (Morton order) within a bidder's reach: geometric (price floor 0) vs with the block's minimum price
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
static unsigned long long s = 88172645463325252ull;
static float rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)); }
static unsigned part(unsigned x) { x &= 1023; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249; return x; }
static float *P2; static int cmp(const void *a, const void *b) { int i = *(const int *)a, j = *(const int *)b; unsigned ci = part(P2[3*i]*1023)|(part(P2[3*i+1]*1023)<<1)|(part(P2[3*i+2]*1023)<<2), cj = part(P2[3*j]*1023)|(part(P2[3*j+1]*1023)<<1)|(part(P2[3*j+2]*1023)<<2); return ci < cj ? -1 : ci > cj; }
int main(int argc, char **argv) {
  int n = 16384, iters = 50, surface = argc > 1 ? atoi(argv[1]) : 0; float eps = 0.005f;
  float *p1 = malloc(n * 12), *p2 = malloc(n * 12), *price = calloc(n, 4), *inc = calloc(n, 4), *max_inc = calloc(n, 4), *val = malloc(n * 4);
  int *assign = malloc(n * 4), *inv = malloc(n * 4), *bid = calloc(n, 4), *max_idx = calloc(n, 4), *un = malloc(n * 4), *perm = malloc(n * 4);
  int *f1 = malloc(n * 4), *f2 = malloc(n * 4);
  for (int i = 0; i < n * 3; ++i) { p1[i] = rnd(); p2[i] = rnd(); }
  if (surface) for (int i = 0; i < n; ++i) for (int w = 0; w < 2; ++w) { float *p = (w ? p2 : p1) + 3 * i; float x, y, z, r; do { x = 2 * rnd() - 1; y = 2 * rnd() - 1; z = 2 * rnd() - 1; r = x * x + y * y + z * z; } while (r > 1 || r < 1e-4); r = 0.5f / sqrtf(r); p[0] = 0.5f + x * r; p[1] = 0.5f + y * r; p[2] = 0.5f + z * r; }
  P2 = p2; for (int i = 0; i < n; ++i) perm[i] = i; qsort(perm, n, 4, cmp);
  int nb = n / 16; float *lo = malloc(nb * 12), *hi = malloc(nb * 12), *bmin = malloc(nb * 4);
  for (int b = 0; b < nb; ++b) for (int a = 0; a < 3; ++a) { float l = 1e9, h = -1e9; for (int c = 0; c < 16; ++c) { float v = p2[3 * perm[16 * b + c] + a]; if (v < l) l = v; if (v > h) h = v; } lo[3 * b + a] = l; hi[3 * b + a] = h; }
  for (int j = 0; j < n; ++j) { assign[j] = inv[j] = -1; f1[j] = f2[j] = -1; }
  for (int it = 0; it < iters; ++it) {
    int cnt = 0; for (int j = 0; j < n; ++j) if (assign[j] == -1) un[cnt++] = j;
    for (int b = 0; b < nb; ++b) { float m = 1e9; for (int c = 0; c < 16; ++c) if (price[perm[16 * b + c]] < m) m = price[perm[16 * b + c]]; bmin[b] = m; }
    double g = 0, pa_ = 0, zero = 0; int have = 0;
    for (int b = 0; b < nb; ++b) zero += bmin[b] == 0.f;
    for (int u = 0; u < cnt; ++u) {
      int j = un[u]; float x = p1[3 * j], y = p1[3 * j + 1], z = p1[3 * j + 2];
      float b1 = -1e9f, b2 = -1e9f; int i1 = -1, i2 = -1;
      for (int k = 0; k < n; ++k) { float dx = p2[3 * k] - x, dy = p2[3 * k + 1] - y, dz = p2[3 * k + 2] - z; float d = (float)((3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz)) - (double)price[k]); val[k] = d;
        if (d > b1) { b2 = b1; i2 = i1; b1 = d; i1 = k; } else if (d > b2) { b2 = d; i2 = k; } }
      if (f1[j] >= 0 && f2[j] >= 0) {
        float va = val[f1[j]], vb = val[f2[j]]; float cm = va < vb ? va : vb; have++;
        for (int b = 0; b < nb; ++b) { float gx = fmaxf(fmaxf(lo[3*b] - x, x - hi[3*b]), 0), gy = fmaxf(fmaxf(lo[3*b+1] - y, y - hi[3*b+1]), 0), gz = fmaxf(fmaxf(lo[3*b+2] - z, z - hi[3*b+2]), 0); float gap = sqrtf(gx*gx+gy*gy+gz*gz);
          if (gap <= 3.0f - cm) g++; if (gap <= 3.0f - cm - bmin[b]) pa_++; }
      }
      f1[j] = i1; f2[j] = i2; bid[j] = i1; inc[j] = b1 - b2 + eps;
    }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; if (inc[j] > max_inc[bid[j]]) max_inc[bid[j]] = inc[j]; }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; float b = inc[j], mi = max_inc[bid[j]]; if (b - 1e-6 <= mi && mi <= b + 1e-6) max_idx[bid[j]] = j; }
    int last = it == iters - 1;
    for (int u = 0; u < cnt; ++u) { int j = un[u], t = bid[j]; if (last || max_idx[t] == j) { int iv = inv[t]; if (!last && iv != -1) assign[iv] = -1; inv[t] = j; assign[j] = t; price[t] += inc[j]; max_inc[t] = -1e9f; } }
    if (have && (it % 5 == 4 || it < 3)) printf("it %2d unassigned %5d: blocks within reach: geometric %.1f, with the block's minimum price %.1f  (blocks holding a zero-price target: %.0f of %d)\n", it, cnt, g / have, pa_ / have, zero, nb);
  }
  return 0;
}
