// emu_mds.cpp -- runs the reference minimum_density_sampling_kernel<1> (text included
// from a file the generator extracts from /root/reference/cuda/MDS/MDS_cuda.cu:81-211 at
// run time) with ONE thread under simt.h.  Only the single-thread instantiation is
// race-free (the multi-thread kernel has two formal data races, see oracle/mds.c), so
// this pins the accumulate / weight / pick / 1e9 logic with bs = 1 (ties -> lowest k).
// `exp(float)` is given its float overload, as CUDA device code resolves it.
// usage: emu_mds in.bin out.bin  (in: int b,n,m; mean_mst_length[b]; xyz)
#include "simt.h"
static inline float exp(float x) { return expf(x); }
#include REF_KERNELS_INC

int main(int argc, char **argv) {
  std::vector<char> in;
  read_all(argv[1], in);
  const int *hdr = reinterpret_cast<const int *>(in.data());
  const int b = hdr[0], n = hdr[1], m = hdr[2];
  float *mml = const_cast<float *>(reinterpret_cast<const float *>(hdr + 3));
  const float *xyz = mml + b;
  std::vector<float> temp((size_t)b * n, 0.f);
  std::vector<int> idx((size_t)b * m, 0);
  simt_launch(minimum_density_sampling_kernel<1>, dim3(b), dim3(1), b, n, m, xyz, temp.data(),
              idx.data(), mml);
  FILE *fo = fopen(argv[2], "wb");
  fwrite(idx.data(), 4, idx.size(), fo);
  fclose(fo);
  return 0;
}
