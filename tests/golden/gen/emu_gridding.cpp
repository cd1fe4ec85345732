// emu_gridding.cpp -- runs the reference gridding / gridding-reverse kernels (text
// included from files the generator extracts at run time from
// /root/reference/cuda/gridding/{gridding.cu,gridding_reverse.cu}) under simt.h.
// Launch shapes restated from gridding.cu:179-211, :314-335 and
// gridding_reverse.cu:105-122, :216-236 (one block per sample, <= 512 threads).
// usage: emu_gridding in.bin out.bin
//   in: int b,npts,scale ; ptcloud[b,npts,3] (already scaled) ; grad_grid[b,scale^3] ;
//       rev_grid[b,scale^3] ; rev_grad_ptcloud[b,scale^3,3]
#include "simt.h"
using std::abs;
#define EPS 1e-6
#include REF_GRIDDING_INC
#include REF_GRIDDING_GRAD_INC
#include REF_REVERSE_INC
#include REF_REVERSE_GRAD_INC

int main(int argc, char **argv) {
  std::vector<char> in;
  read_all(argv[1], in);
  const int *hdr = reinterpret_cast<const int *>(in.data());
  const int b = hdr[0], npts = hdr[1], scale = hdr[2];
  const int s = scale / 2, len = 2 * s, nv = len * len * len, n3 = scale * scale * scale;
  const float *pt = reinterpret_cast<const float *>(hdr + 3);
  const float *grad_grid = pt + (size_t)b * npts * 3;
  const float *rev_grid = grad_grid + (size_t)b * nv;
  const float *rev_gp = rev_grid + (size_t)b * n3;
  const int threads = 64;  // any block size is valid: the kernels are block-stride loops
  std::vector<float> grid((size_t)b * nv, 0.f), w((size_t)b * npts * 24, 0.f), gpt((size_t)b * npts * 3, 0.f);
  std::vector<int> ix((size_t)b * npts * 8, 0);
  simt_launch(gridding_kernel, dim3(b), dim3(threads), nv, npts, (float)-s, (float)-s, (float)-s, len, len,
              pt, grid.data(), w.data(), ix.data());
  simt_launch(gridding_grad_kernel, dim3(b), dim3(threads), nv, npts, (const float *)w.data(),
              (const int *)ix.data(), grad_grid, gpt.data());
  std::vector<float> rpt((size_t)b * n3 * 3, 0.f), rgg((size_t)b * n3, 0.f);
  simt_launch(gridding_reverse_kernel, dim3(b), dim3(threads), scale, n3, rev_grid, rpt.data());
  simt_launch(gridding_reverse_grad_kernel, dim3(b), dim3(threads), scale, n3, (const float *)rpt.data(),
              rev_grid, rev_gp, rgg.data());
  FILE *fo = fopen(argv[2], "wb");
  fwrite(grid.data(), 4, grid.size(), fo);
  fwrite(w.data(), 4, w.size(), fo);
  fwrite(ix.data(), 4, ix.size(), fo);
  fwrite(gpt.data(), 4, gpt.size(), fo);
  fwrite(rpt.data(), 4, rpt.size(), fo);
  fwrite(rgg.data(), 4, rgg.size(), fo);
  fclose(fo);
  return 0;
}
