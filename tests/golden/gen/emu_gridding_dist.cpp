// emu_gridding_dist.cpp -- runs the reference gridding-distance kernels (text included from
// files the generator extracts at run time from
// /root/reference/cuda/gridding_loss/gridding_distance.cu) under simt.h.
// Launch shape restated from gridding_distance.cu:180-211, :307-329 (one block per sample).
// usage: emu_gridding_dist in.bin out.bin
//   in: int b,npts,min_x,max_x,min_y,max_y,min_z,max_z ; ptcloud[b,npts,3] ; grad_grid[b,nverts*8]
#include "simt.h"
using std::abs;
#include REF_GDIST_INC
#include REF_GDIST_GRAD_INC

int main(int argc, char **argv) {
  std::vector<char> in;
  read_all(argv[1], in);
  const int *hdr = reinterpret_cast<const int *>(in.data());
  const int b = hdr[0], npts = hdr[1];
  const int min_x = hdr[2], max_x = hdr[3], min_y = hdr[4], max_y = hdr[5], min_z = hdr[6], max_z = hdr[7];
  const int len_x = max_x - min_x + 1, len_y = max_y - min_y + 1, len_z = max_z - min_z + 1;
  const int nv = len_x * len_y * len_z;
  const float *pt = reinterpret_cast<const float *>(hdr + 8);
  const float *grad_grid = pt + (size_t)b * npts * 3;
  const int threads = 64;  // any block size is valid: the kernels are block-stride loops
  std::vector<float> grid((size_t)b * nv * 8, 0.f), w((size_t)b * npts * 24, 0.f), gpt((size_t)b * npts * 3, 0.f);
  std::vector<int> ix((size_t)b * npts * 8, 0);
  simt_launch(gridding_dist_kernel, dim3(b), dim3(threads), nv, npts, (float)min_x, (float)min_y,
              (float)min_z, len_y, len_z, pt, grid.data(), w.data(), ix.data());
  simt_launch(gridding_dist_grad_kernel, dim3(b), dim3(threads), nv, npts, (const float *)w.data(),
              (const int *)ix.data(), grad_grid, gpt.data());
  FILE *fo = fopen(argv[2], "wb");
  fwrite(grid.data(), 4, grid.size(), fo);
  fwrite(w.data(), 4, w.size(), fo);
  fwrite(ix.data(), 4, ix.size(), fo);
  fwrite(gpt.data(), 4, gpt.size(), fo);
  fclose(fo);
  return 0;
}
