// emu_cubic.cpp -- runs the reference cubic_feature_sampling kernels (text included from
// files the generator extracts at run time from
// /root/reference/cuda/cubic_feature_sampling/cubic_feature_sampling.cu) under simt.h.
// Launch shape restated from cubic_feature_sampling.cu:104-133, :176-205.
// usage: emu_cubic in.bin out.bin
//   in: int b,npts,c,scale,ns ; ptcloud[b,npts,3] (voxel space) ; feat[b,c,scale^3] ;
//       grad_out[b,npts,(2ns)^3,c]
#include "simt.h"
#include REF_CUBIC_INC
#include REF_CUBIC_GRAD_INC

int main(int argc, char **argv) {
  std::vector<char> in;
  read_all(argv[1], in);
  const int *hdr = reinterpret_cast<const int *>(in.data());
  const int b = hdr[0], npts = hdr[1], c = hdr[2], scale = hdr[3], ns = hdr[4];
  const int nv = 8 * ns * ns * ns, cub = scale * scale * scale;
  const float *pt = reinterpret_cast<const float *>(hdr + 5);
  const float *feat = pt + (size_t)b * npts * 3;
  const float *gout = feat + (size_t)b * c * cub;
  std::vector<float> out((size_t)b * npts * nv * c, 0.f), gpt((size_t)b * npts * 3, 0.f),
      gfeat((size_t)b * c * cub, 0.f);
  std::vector<int> ix((size_t)b * npts * nv, 0);
  simt_launch(cubic_feature_sampling_kernel, dim3(b), dim3(32), scale, ns, nv, npts, c, pt, feat,
              out.data(), ix.data());
  simt_launch(cubic_feature_sampling_grad_kernel, dim3(b), dim3(32), scale, ns, nv, npts, c, gout,
              (const int *)ix.data(), gpt.data(), gfeat.data());
  FILE *fo = fopen(argv[2], "wb");
  fwrite(out.data(), 4, out.size(), fo);
  fwrite(ix.data(), 4, ix.size(), fo);
  fwrite(gfeat.data(), 4, gfeat.size(), fo);
  fwrite(gpt.data(), 4, gpt.size(), fo);
  fclose(fo);
  return 0;
}
