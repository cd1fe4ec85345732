// emu_emd.cpp -- runs the reference EMD kernels (text included from a file the
// generator extracts from /root/reference/cuda/emd/emd_cuda.cu at run time)
// under simt.h.  Host loop restated from emd_cuda.cu:256-269; scratch
// initialisation from cuda/emd/emd_module.py:43-54.
// usage: emu_emd in.bin out.bin   (in: int b,n,iters; float eps; xyz1; xyz2)
#include "simt.h"
#include REF_KERNELS_INC

int main(int argc, char **argv) {
  std::vector<char> in;
  read_all(argv[1], in);
  const int *hdr = reinterpret_cast<const int *>(in.data());
  const int b = hdr[0], n = hdr[1], iters = hdr[2];
  float eps;
  std::memcpy(&eps, hdr + 3, 4);
  const float *xyz1 = reinterpret_cast<const float *>(hdr + 4);
  const float *xyz2 = xyz1 + (size_t)b * n * 3;
  std::vector<float> dist(b * n, 0.f), price(b * n, 0.f), bid_inc(b * n, 0.f), max_inc(b * n, 0.f);
  std::vector<int> assignment(b * n, -1), assignment_inv(b * n, -1), bid(b * n, 0),
      unass_idx(b * n, 0), max_idx(b * n, 0), unass_cnt(512, 0), unass_cnt_sum(512, 0),
      cnt_tmp(512, 0);
  std::vector<int> trace(iters, 0);
  FILE *fo = fopen(argv[2], "wb");
  for (int i = 0; i < iters; i++) {
    simt_launch(clear, dim3(1), dim3(b), b, cnt_tmp.data(), unass_cnt.data());
    simt_launch(calc_unass_cnt, dim3(b, n / 1024, 1), dim3(1024), b, n, assignment.data(), unass_cnt.data());
    simt_launch(calc_unass_cnt_sum, dim3(1), dim3(b), b, unass_cnt.data(), unass_cnt_sum.data());
    simt_launch(calc_unass_idx, dim3(b, n / 1024, 1), dim3(1024), b, n, assignment.data(),
                unass_idx.data(), unass_cnt.data(), unass_cnt_sum.data(), cnt_tmp.data());
    for (int q = 0; q < b; ++q) trace[i] += unass_cnt[q];
    simt_launch(Bid, dim3(b, n / 1024, 1), dim3(1024), b, n, xyz1, xyz2, eps, assignment.data(),
                assignment_inv.data(), price.data(), bid.data(), bid_inc.data(), max_inc.data(),
                unass_cnt.data(), unass_cnt_sum.data(), unass_idx.data());
    simt_launch(GetMax, dim3(b, n / 1024, 1), dim3(1024), b, n, assignment.data(), bid.data(),
                bid_inc.data(), max_inc.data(), max_idx.data());
    simt_launch(Assign, dim3(b, n / 1024, 1), dim3(1024), b, n, assignment.data(),
                assignment_inv.data(), price.data(), bid.data(), bid_inc.data(), max_inc.data(),
                max_idx.data(), i == iters - 1);
  }
  simt_launch(CalcDist, dim3(b, n / 1024, 1), dim3(1024), b, n, const_cast<float *>(xyz1),
              const_cast<float *>(xyz2), dist.data(), assignment.data());
  fwrite(dist.data(), 4, dist.size(), fo);
  fwrite(assignment.data(), 4, assignment.size(), fo);
  fwrite(price.data(), 4, price.size(), fo);
  fwrite(trace.data(), 4, trace.size(), fo);
  fclose(fo);
  return 0;
}
