// ref_p2i.cpp -- drives the reference's own p2i functors on the CPU to produce
// golden vectors.  The functor text (p2i_max.h:7-143, p2i_sum.h:7-131) is pulled
// at run time from /root/reference/cuda/p2i_op by tests/golden/gen_p2i.py and
// included below; utility.h / common.h are included where they lie.  Two
// accommodations, both in THIS file: (1) only the functor structs are included,
// not the op wrappers (their AT_DISPATCH_FLOATING_TYPES(points.type(), ...) no
// longer compiles against current torch); (2) a correct non-template CPU
// atomic_cas overload -- the reference's CPU atomic_cas (utility.h:35-42) returns
// the NEW value, so its forward functor spins forever on the CPU.
// usage: ref_p2i in.bin out.bin
//   in : int npoints,channels,batch,h,w; float radius; points; feat; batch_inds; background; out_grad
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "utility.h"
namespace haya_ext {
inline int32_t atomic_cas(int32_t *a, int32_t c, int32_t v) {
  int32_t o = *a;
  if (o == c) *a = v;
  return o;
}
}  // namespace haya_ext
#include REF_MAX_INC
#include REF_SUM_INC
using namespace haya_ext;

int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  int hdr[5];
  float radius;
  fread(hdr, 4, 5, f);
  fread(&radius, 4, 1, f);
  const int n = hdr[0], C = hdr[1], B = hdr[2], H = hdr[3], W = hdr[4];
  std::vector<float> points(n * 2), feat(n * C), bg((size_t)B * C * H * W), og(bg.size());
  std::vector<int32_t> bi(n);
  fread(points.data(), 4, points.size(), f);
  fread(feat.data(), 4, feat.size(), f);
  fread(bi.data(), 4, bi.size(), f);
  fread(bg.data(), 4, bg.size(), f);
  fread(og.data(), 4, og.size(), f);
  fclose(f);
  const size_t px = bg.size();
  // ---- max forward / backward
  std::vector<float> out(bg);
  std::vector<int32_t> ids(px, -1), lock(px, 0);
  kernel<cpu_device>::launch(p2i_max_forward_kernel<float>(), n * C, points.data(), feat.data(),
                             bi.data(), out.data(), ids.data(), lock.data(), B, n, C, 0, radius, H, W);
  std::vector<float> gp(n * 2, 0.f), gf(n * C, 0.f), gb(px, 0.f);
  kernel<cpu_device>::launch(p2i_max_backward_kernel<float>(), (int)px, og.data(), ids.data(),
                             points.data(), feat.data(), gp.data(), gf.data(), gb.data(), B, n, C, 0,
                             radius, H, W);
  // ---- sum forward / backward
  std::vector<float> sout(bg);
  std::fill(lock.begin(), lock.end(), 0);
  kernel<cpu_device>::launch(p2i_sum_forward_kernel<float>(), n * C, points.data(), feat.data(),
                             bi.data(), sout.data(), lock.data(), B, n, C, 0, radius, H, W);
  std::vector<float> sgp(n * 2, 0.f), sgf(n * C, 0.f);
  kernel<cpu_device>::launch(p2i_sum_backward_kernel<float>(), n * C, og.data(), points.data(),
                             feat.data(), bi.data(), sgp.data(), sgf.data(), B, n, C, 0, radius, H, W);
  FILE *o = fopen(argv[2], "wb");
  fwrite(out.data(), 4, px, o);
  fwrite(ids.data(), 4, px, o);
  fwrite(gp.data(), 4, gp.size(), o);
  fwrite(gf.data(), 4, gf.size(), o);
  fwrite(gb.data(), 4, px, o);
  fwrite(sout.data(), 4, px, o);
  fwrite(sgp.data(), 4, sgp.size(), o);
  fwrite(sgf.data(), 4, sgf.size(), o);
  fclose(o);
  return 0;
}
