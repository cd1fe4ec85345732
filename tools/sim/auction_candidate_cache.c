// auction with a per-bidder candidate cache: how often does a late-iteration bidder need a full search?
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
static unsigned long long s = 88172645463325252ull;
static float rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)); }
int main(int argc, char **argv) {
  int n = 16384, iters = 50, K = argc > 1 ? atoi(argv[1]) : 8; float delta = argc > 2 ? atof(argv[2]) : 0.02f, eps = 0.005f;
  int surface = argc > 3 ? atoi(argv[3]) : 0;
  float *p1 = malloc(n * 12), *p2 = malloc(n * 12), *price = calloc(n, 4), *inc = calloc(n, 4), *max_inc = calloc(n, 4), *val = malloc(n * 4);
  int *assign = malloc(n * 4), *inv = malloc(n * 4), *bid = calloc(n, 4), *max_idx = calloc(n, 4), *un = malloc(n * 4);
  int *cidx = malloc((size_t)n * K * 4), *ccnt = calloc(n, 4); float *U = malloc(n * 4);
  for (int i = 0; i < n * 3; ++i) { p1[i] = rnd(); p2[i] = rnd(); }
  if (surface) for (int i = 0; i < n; ++i) { // points on a sphere, two different samplings
    for (int w = 0; w < 2; ++w) { float *p = (w ? p2 : p1) + 3 * i; float x, y, z, r; do { x = 2 * rnd() - 1; y = 2 * rnd() - 1; z = 2 * rnd() - 1; r = x * x + y * y + z * z; } while (r > 1 || r < 1e-4); r = 0.5f / sqrtf(r); p[0] = 0.5f + x * r; p[1] = 0.5f + y * r; p[2] = 0.5f + z * r; } }
  for (int j = 0; j < n; ++j) assign[j] = inv[j] = -1;
  long long tot_eval = 0;
  for (int it = 0; it < iters; ++it) {
    int cnt = 0; for (int j = 0; j < n; ++j) if (assign[j] == -1) un[cnt++] = j;
    int miss = 0, bad = 0; long long cached_sizes = 0;
    for (int u = 0; u < cnt; ++u) {
      int j = un[u]; float x = p1[3 * j], y = p1[3 * j + 1], z = p1[3 * j + 2];
      // truth
      float best = -1e9f, better = -1e9f; int bi = -1;
      for (int k = 0; k < n; ++k) { float dx = p2[3 * k] - x, dy = p2[3 * k + 1] - y, dz = p2[3 * k + 2] - z; float d = (float)((3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz)) - (double)price[k]); val[k] = d; if (d > best) { better = best; best = d; bi = k; } else if (d > better) better = d; }
      int hit = 0;
      if (ccnt[j] >= 2) { float b1 = -1e9f, b2 = -1e9f; int i1 = -1; for (int c = 0; c < ccnt[j]; ++c) { int k = cidx[(size_t)j * K + c]; float d = val[k]; if (d > b1) { b2 = b1; b1 = d; i1 = k; } else if (d > b2) b2 = d; }
        if (b2 > U[j]) { hit = 1; if (b1 != best || b2 != better) bad++; } }
      if (!hit) { miss++; // rebuild: all k with val >= better - delta, top K
        float thr = better - delta; int c = 0; float kth = thr; // simple selection
        // collect candidates
        static int cand[16384]; int nc = 0; for (int k = 0; k < n; ++k) if (val[k] >= thr) cand[nc++] = k;
        // partial sort by value desc
        for (int a = 0; a < nc && a <= K; ++a) { int m = a; for (int b2_ = a + 1; b2_ < nc; ++b2_) if (val[cand[b2_]] > val[cand[m]]) m = b2_; int t = cand[a]; cand[a] = cand[m]; cand[m] = t; }
        c = nc < K ? nc : K; for (int a = 0; a < c; ++a) cidx[(size_t)j * K + a] = cand[a]; ccnt[j] = c; U[j] = nc > K ? val[cand[K]] : thr; (void)kth; tot_eval += nc; }
      cached_sizes += ccnt[j];
      bid[j] = bi; inc[j] = best - better + eps;
    }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; if (inc[j] > max_inc[bid[j]]) max_inc[bid[j]] = inc[j]; }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; float b = inc[j], mi = max_inc[bid[j]]; if (b - 1e-6 <= mi && mi <= b + 1e-6) max_idx[bid[j]] = j; }
    int last = it == iters - 1;
    for (int u = 0; u < cnt; ++u) { int j = un[u], t = bid[j]; if (last || max_idx[t] == j) { int iv = inv[t]; if (!last && iv != -1) assign[iv] = -1; inv[t] = j; assign[j] = t; price[t] += inc[j]; max_inc[t] = -1e9f; } }
    printf("it %2d unassigned %5d full searches %5d (%.1f%%) bad %d mean cache %.1f\n", it, cnt, miss, 100.0 * miss / (cnt ? cnt : 1), bad, (double)cached_sizes / (cnt ? cnt : 1));
  }
  return 0;
}
