// emu_expansion.cpp -- runs the reference calc_penalty kernel (text included from
// a file the generator extracts from
// /root/reference/cuda/expansion_penalty/expansion_penalty_cuda.cu at run time)
// under simt.h.  Launch shape restated from expansion_penalty_cuda.cu:151-157,
// scratch from expansion_penalty_module.py:31-35.
// usage: emu_expansion in.bin out.bin (in: int b,n,P; float alpha; xyz)
#include "simt.h"
#include REF_KERNELS_INC

int main(int argc, char **argv) {
  std::vector<char> in;
  read_all(argv[1], in);
  const int *hdr = reinterpret_cast<const int *>(in.data());
  const int b = hdr[0], n = hdr[1], P = hdr[2];
  float alpha;
  std::memcpy(&alpha, hdr + 3, 4);
  const float *xyz = reinterpret_cast<const float *>(hdr + 4);
  std::vector<float> dist(b * n, 0.f), cost((size_t)b * n * 512, 0.f), mean(b, 0.f);
  std::vector<int> idx(b * n, -1), neighbor((size_t)b * n * 512, 0);
  simt_launch(calc_penalty, dim3(b, n / P, 1), dim3(P), b, n, P, xyz, idx.data(), dist.data(),
              alpha, neighbor.data(), cost.data(), mean.data());
  FILE *fo = fopen(argv[2], "wb");
  fwrite(dist.data(), 4, dist.size(), fo);
  fwrite(idx.data(), 4, idx.size(), fo);
  fwrite(mean.data(), 4, mean.size(), fo);  // un-normalised sum over patches
  fclose(fo);
  return 0;
}
