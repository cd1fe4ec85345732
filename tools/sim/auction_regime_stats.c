// What the data does to the auction, per iteration (plain float arithmetic, statistics only): unassigned bidders,
// 16-target blocks (Morton order) within a bidder's reach with the price floor 0 (what bid_scan tests today) and with
// the block's minimum price, targets that pass the precise filter with the bound from the two previous favourites,
// the length of the targets' bidder lists (what the award phase's walker follows) and how many entries remain when a
// bidder that is already outbid at the moment it arrives does not link itself.
//   gcc -O2 -o /tmp/ars tools/sim/auction_regime_stats.c -lm && /tmp/ars clouds.bin   (clouds.bin: x[n*3], y[n*3] float32, n = 16384)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static unsigned part(unsigned x) { x &= 1023; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249; return x; }
static float *P2, LO[3], SC[3];
static unsigned code(int i) { return part((unsigned)((P2[3*i]-LO[0])*SC[0])) | (part((unsigned)((P2[3*i+1]-LO[1])*SC[1])) << 1) | (part((unsigned)((P2[3*i+2]-LO[2])*SC[2])) << 2); }
static int cmp(const void *a, const void *b) { unsigned ci = code(*(const int *)a), cj = code(*(const int *)b); return ci < cj ? -1 : ci > cj; }
static unsigned long long s = 88172645463325252ull;
static unsigned rnd32() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 20); }
int main(int argc, char **argv) {
  int n = 16384, iters = 50; float eps = 0.005f;
  FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
  float *p1 = malloc(n * 12), *p2 = malloc(n * 12);
  if (fread(p1, 12, n, f) != (size_t)n || fread(p2, 12, n, f) != (size_t)n) return 2;
  float *price = calloc(n, 4), *inc = calloc(n, 4), *max_inc = calloc(n, 4), *val = malloc(n * 4);
  int *assign = malloc(n * 4), *inv = malloc(n * 4), *bid = calloc(n, 4), *max_idx = calloc(n, 4), *un = malloc(n * 4), *perm = malloc(n * 4);
  int *f1 = malloc(n * 4), *f2 = malloc(n * 4), *cnt_t = malloc(n * 4), *ord = malloc(n * 4); float *runmax = malloc(n * 4); int *kept = malloc(n * 4);
  for (int a = 0; a < 3; ++a) { float l = 1e9, h = -1e9; for (int i = 0; i < n; ++i) { if (p2[3*i+a] < l) l = p2[3*i+a]; if (p2[3*i+a] > h) h = p2[3*i+a]; } LO[a] = l; SC[a] = 1023.0f / (h - l + 1e-9f); }
  P2 = p2; for (int i = 0; i < n; ++i) perm[i] = i; qsort(perm, n, 4, cmp);
  int nb = n / 16; float *lo = malloc(nb * 12), *hi = malloc(nb * 12), *bmin = malloc(nb * 4);
  for (int b = 0; b < nb; ++b) for (int a = 0; a < 3; ++a) { float l = 1e9, h = -1e9; for (int c = 0; c < 16; ++c) { float v = p2[3 * perm[16 * b + c] + a]; if (v < l) l = v; if (v > h) h = v; } lo[3 * b + a] = l; hi[3 * b + a] = h; }
  for (int j = 0; j < n; ++j) { assign[j] = inv[j] = -1; f1[j] = f2[j] = -1; }
  for (int it = 0; it < iters; ++it) {
    int cnt = 0; for (int j = 0; j < n; ++j) if (assign[j] == -1) un[cnt++] = j;
    for (int b = 0; b < nb; ++b) { float m = 1e9; for (int c = 0; c < 16; ++c) if (price[perm[16 * b + c]] < m) m = price[perm[16 * b + c]]; bmin[b] = m; }
    double g = 0, pa_ = 0, passes = 0, passes_true = 0; int have = 0; double gmax = 0, pmax = 0;
    for (int u = 0; u < cnt; ++u) {
      int j = un[u]; float x = p1[3 * j], y = p1[3 * j + 1], z = p1[3 * j + 2];
      float b1 = -1e9f, b2 = -1e9f; int i1 = -1, i2 = -1;
      for (int k = 0; k < n; ++k) { float dx = p2[3 * k] - x, dy = p2[3 * k + 1] - y, dz = p2[3 * k + 2] - z; float d = (float)((3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz)) - (double)price[k]); val[k] = d;
        if (d > b1) { b2 = b1; i2 = i1; b1 = d; i1 = k; } else if (d > b2) { b2 = d; i2 = k; } }
      if (f1[j] >= 0 && f2[j] >= 0) {
        float va = val[f1[j]], vb = val[f2[j]]; float cm = va < vb ? va : vb; have++;
        double gj = 0, pj = 0;
        for (int b = 0; b < nb; ++b) { float gx = fmaxf(fmaxf(lo[3*b] - x, x - hi[3*b]), 0), gy = fmaxf(fmaxf(lo[3*b+1] - y, y - hi[3*b+1]), 0), gz = fmaxf(fmaxf(lo[3*b+2] - z, z - hi[3*b+2]), 0); float gap = sqrtf(gx*gx+gy*gy+gz*gz);
          if (gap <= 3.0f - cm) gj++; if (gap <= 3.0f - cm - bmin[b]) pj++; }
        g += gj; pa_ += pj; if (gj > gmax) gmax = gj; if (pj > pmax) pmax = pj;
        for (int k = 0; k < n; ++k) { if (val[k] >= cm) passes++; if (val[k] >= b2) passes_true++; }
      }
      f1[j] = i1; f2[j] = i2; bid[j] = i1; inc[j] = b1 - b2 + eps;
    }
    // list lengths; arrivals in random order, an arrival already below the running maximum by more than 1e-6 does not link
    memset(cnt_t, 0, n * 4); memset(kept, 0, n * 4); for (int k = 0; k < n; ++k) runmax[k] = max_inc[k];
    for (int u = 0; u < cnt; ++u) ord[u] = un[u];
    for (int u = cnt - 1; u > 0; --u) { int r = rnd32() % (u + 1); int t = ord[u]; ord[u] = ord[r]; ord[r] = t; }
    for (int u = 0; u < cnt; ++u) { int j = ord[u], t = bid[j]; cnt_t[t]++; if (!((double)runmax[t] > (double)inc[j] + 1e-6)) kept[t]++; if (inc[j] > runmax[t]) runmax[t] = inc[j]; }
    int lmax = 0, kmax = 0, tgts = 0; double ksum = 0; for (int k = 0; k < n; ++k) { if (cnt_t[k]) tgts++; if (cnt_t[k] > lmax) lmax = cnt_t[k]; if (kept[k] > kmax) kmax = kept[k]; ksum += kept[k]; }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; if (inc[j] > max_inc[bid[j]]) max_inc[bid[j]] = inc[j]; }
    for (int u = 0; u < cnt; ++u) { int j = un[u]; float b = inc[j], mi = max_inc[bid[j]]; if (b - 1e-6 <= mi && mi <= b + 1e-6) max_idx[bid[j]] = j; }
    int last = it == iters - 1;
    for (int u = 0; u < cnt; ++u) { int j = un[u], t = bid[j]; if (last || max_idx[t] == j) { int iv = inv[t]; if (!last && iv != -1) assign[iv] = -1; inv[t] = j; assign[j] = t; price[t] += inc[j]; max_inc[t] = -1e9f; } }
    if (it < 4 || it % 5 == 4)
      printf("it %2d unassigned %5d | blocks in reach: floor-0 %.0f (max %.0f), block-min-price %.0f (max %.0f) of %d | filter passes/bidder: fav bound %.1f, final bound %.1f | targets bid on %d, longest list %d, linked after outbid-skip: longest %d, total %.0f\n",
             it, cnt, have ? g / have : 0, gmax, have ? pa_ / have : 0, pmax, nb, have ? passes / have : 0, have ? passes_true / have : 0, tgts, lmax, kmax, ksum);
  }
  return 0;
}
