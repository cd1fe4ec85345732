/*
 * sn_expf.h -- the exponential used by minimum density sampling.
 *
 * Numeric contract shared VERBATIM by the HIP kernel (sparenet_amd/csrc/mds.hip)
 * and the CPU oracle (oracle/mds.c): it is built only from fmaf, one multiply,
 * adds and integer bit manipulation, all correctly rounded IEEE operations, so
 * both sides return bit-identical results.  (CUDA's expf, glibc's expf and ROCm's
 * OCML expf differ from each other in the last ulp; the reference's MDS arg-min
 * turns such a difference into a different sample sequence, so index parity with
 * a libm-based oracle is not attainable on any GPU.  See DESIGN.md, "MDS".)
 *
 * Accuracy: Cephes-style range reduction + degree-5 polynomial, <= 2 ulp on
 * [-104, 88]; subnormal results are produced (two-step scaling), x < -104
 * (below 2^-150) and NaN return 0.
 */
#ifndef SN_EXPF_H
#define SN_EXPF_H

#ifdef __HIPCC__
#define SN_EXPF_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define SN_EXPF_HD static inline
#endif

SN_EXPF_HD float sn_bits_to_float(int bits) {
#ifdef __HIP_DEVICE_COMPILE__
  return __int_as_float(bits);
#else
  float f;
  memcpy(&f, &bits, 4);
  return f;
#endif
}

SN_EXPF_HD float sn_expf(float x) {
  if (!(x > -104.0f)) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  /* n = round-to-nearest-even(x * log2(e)) through the 1.5*2^23 magic constant */
  const float magic = 12582912.0f;
  const float t = fmaf(x, 1.44269504088896341f, magic);
  const float fn = t - magic;
  const int n = (int)fn;
  float r = fmaf(fn, -0.693359375f, x);
  r = fmaf(fn, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  float y = fmaf(p, r2, r) + 1.0f;
  if (n < -126) {
    y = y * sn_bits_to_float((n + 64 + 127) << 23);
    return y * 5.42101086242752217e-20f; /* 2^-64: single rounding into the subnormals */
  }
  return y * sn_bits_to_float((n + 127) << 23);
}

#endif
