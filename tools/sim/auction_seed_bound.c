// how tight is the bound from the previous favourites?  2 seeds (best, second) vs 3 seeds (+ third)
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
static unsigned long long s = 88172645463325252ull;
static float rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)); }
int main(int argc, char **argv) {
  int n = 16384, iters = 50; float eps = 0.005f;
  float *p1 = malloc(n * 12), *p2 = malloc(n * 12), *price = calloc(n, 4), *inc = calloc(n, 4), *max_inc = calloc(n, 4), *val = malloc(n * 4);
  int *assign = malloc(n * 4), *inv = malloc(n * 4), *bid = calloc(n, 4), *max_idx = calloc(n, 4), *un = malloc(n * 4);
  int *f1 = malloc(n * 4), *f2 = malloc(n * 4), *f3 = malloc(n * 4);
  for (int i = 0; i < n * 3; ++i) { p1[i] = rnd(); p2[i] = rnd(); }
  for (int j = 0; j < n; ++j) { assign[j] = inv[j] = -1; f1[j] = f2[j] = f3[j] = -1; }
  for (int it = 0; it < iters; ++it) {
    int cnt = 0; for (int j = 0; j < n; ++j) if (assign[j] == -1) un[cnt++] = j;
    double sum_d2 = 0, sum_d3 = 0, reach2 = 0, reach3 = 0, reachT = 0, ge2 = 0, ge3 = 0; int have = 0;
    for (int u = 0; u < cnt; ++u) {
      int j = un[u]; float x = p1[3 * j], y = p1[3 * j + 1], z = p1[3 * j + 2];
      float b1 = -1e9f, b2 = -1e9f, b3 = -1e9f; int i1 = -1, i2 = -1, i3 = -1;
      for (int k = 0; k < n; ++k) { float dx = p2[3 * k] - x, dy = p2[3 * k + 1] - y, dz = p2[3 * k + 2] - z; float d = (float)((3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz)) - (double)price[k]); val[k] = d;
        if (d > b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; } else if (d > b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; } else if (d > b3) { b3 = d; i3 = k; } }
      if (f1[j] >= 0 && f3[j] >= 0) {
        float va = val[f1[j]], vb = val[f2[j]], vc = val[f3[j]];
        float cm2 = va < vb ? va : vb;
        float hi = va > vb ? va : vb, lo = cm2; float cm3 = vc > hi ? hi : (vc > lo ? vc : lo);  // second largest of three
        sum_d2 += b2 - cm2; sum_d3 += b2 - cm3; have++;
        int r2c = 0, r3c = 0, rT = 0, g2 = 0, g3 = 0;
        for (int k = 0; k < n; ++k) { float dist = 3.0f - val[k] - price[k]; if (dist <= 3.0f - cm2) r2c++; if (dist <= 3.0f - cm3) r3c++; if (dist <= 3.0f - b2) rT++; if (val[k] >= cm2) g2++; if (val[k] >= cm3) g3++; }
        reach2 += r2c; reach3 += r3c; reachT += rT; ge2 += g2; ge3 += g3;
      }
      f1[j] = i1; f2[j] = i2; f3[j] = i3;
      bid[j] = i1; inc[j] = b1 - b2 + eps;
    }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; if (inc[j] > max_inc[bid[j]]) max_inc[bid[j]] = inc[j]; }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; float b = inc[j], mi = max_inc[bid[j]]; if (b - 1e-6 <= mi && mi <= b + 1e-6) max_idx[bid[j]] = j; }
    int last = it == iters - 1;
    for (int u = 0; u < cnt; ++u) { int j = un[u], t = bid[j]; if (last || max_idx[t] == j) { int iv = inv[t]; if (!last && iv != -1) assign[iv] = -1; inv[t] = j; assign[j] = t; price[t] += inc[j]; max_inc[t] = -1e9f; } }
    if (have && (it % 5 == 4 || it < 3)) printf("it %2d unassigned %5d: better - cm: 2 seeds %.4f, 3 seeds %.4f | targets within reach: 2 seeds %.0f, 3 seeds %.0f, ideal %.0f | value >= cm: %.1f / %.1f\n", it, cnt, sum_d2 / have, sum_d3 / have, reach2 / have, reach3 / have, reachT / have, ge2 / have, ge3 / have);
  }
  return 0;
}
