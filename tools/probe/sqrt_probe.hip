#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const float* in, float* o1, float* o2, float* o3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { o1[i] = __fsqrt_rn(in[i]); o2[i] = sqrtf(in[i]); o3[i] = __builtin_sqrtf(in[i]); }
}
int main() {
  const int n = 1 << 20;
  float* h = (float*)malloc(n * 4);
  srand(1);
  h[0] = 0.01764967f; h[1] = 0.023348428f; h[2] = 0.02025844f;
  for (int i = 3; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 3.0f;
  float *d, *o1, *o2, *o3;
  hipMalloc(&d, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4); hipMalloc(&o3, n * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, o1, o2, o3, n);
  float *r1 = (float*)malloc(n * 4), *r2 = (float*)malloc(n * 4), *r3 = (float*)malloc(n * 4);
  hipMemcpy(r1, o1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(r3, o3, n * 4, hipMemcpyDeviceToHost);
  int b1 = 0, b2 = 0, b3 = 0;
  for (int i = 0; i < n; ++i) { float c = sqrtf(h[i]); b1 += (r1[i] != c); b2 += (r2[i] != c); b3 += (r3[i] != c); }
  printf("mismatch vs host sqrtf: __fsqrt_rn %d  sqrtf %d  __builtin_sqrtf %d of %d\n", b1, b2, b3, n);
  for (int i = 0; i < 3; ++i) printf("%.9g: %.9g %.9g %.9g host %.9g\n", h[i], r1[i], r2[i], r3[i], sqrtf(h[i]));
  return 0;
}
