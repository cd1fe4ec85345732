// Exhaustive check of the MDS quotient shortcut (mds.hip, neg_div2): for a fixed divisor t and
// rt = RN(1/t), q = n*rt refined by two exact-residual corrections must equal RN(n/t) for every
// float n = -d, d in [2^-60, 2^9).  Prints the number of mismatching d per t (expected 0).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_probe div_probe.hip && ./div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void probe(float t, unsigned first_bits, unsigned long long count,
                      unsigned long long *bad, unsigned *example) {
#pragma clang fp contract(off)
  const float rt = 1.0f / t;
  unsigned long long local = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float d = __uint_as_float(first_bits + (unsigned)i);
    const float n = -d;
    float q = n * rt;
    float r = __builtin_fmaf(-t, q, n);
    q = __builtin_fmaf(r, rt, q);
    r = __builtin_fmaf(-t, q, n);
    q = __builtin_fmaf(r, rt, q);
    const float ref = n / t;
    if (__float_as_uint(q) != __float_as_uint(ref)) {
      ++local;
      *example = __float_as_uint(d);
    }
  }
  if (local) atomicAdd(bad, local);
}

int main() {
  std::vector<float> ts = {5.0f * 0.0085f * 0.0085f, 0.018f, 1.0f, 1.0f / 3.0f, 1.99999988f,
                           1.00000012f, 1.17e-3f, 7.3e-5f, 0.75f, 1.5f, 1e-9f, 1e6f};
  srand(1234);
  for (int i = 0; i < 52; ++i) {
    unsigned bits = ((unsigned)(100 + rand() % 40) << 23) | ((unsigned)rand() & 0x7fffffu);
    float f;
    memcpy(&f, &bits, 4);
    ts.push_back(f);
  }
  unsigned long long *bad;
  unsigned *ex;
  hipMalloc(&bad, 8);
  hipMalloc(&ex, 4);
  const unsigned first = (127u - 60u) << 23;
  const unsigned long long count = 69ull << 23;
  unsigned long long total = 0;
  for (float t : ts) {
    hipMemset(bad, 0, 8);
    hipMemset(ex, 0, 4);
    probe<<<4096, 256>>>(t, first, count, bad, ex);
    unsigned long long h;
    unsigned e;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&e, ex, 4, hipMemcpyDeviceToHost);
    total += h;
    if (h) printf("t=%.9g mismatches=%llu example d bits=0x%08x\n", t, h, e);
  }
  printf("div_probe: %zu divisors x %llu dividends, mismatches=%llu\n", ts.size(), count, total);
  return total != 0;
}
