// traffic_probe.hip -- known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the
// access kinds the persistent auction uses: 4 / 8-byte relaxed agent-scope ("sc1", coherent) loads and stores,
// next to plain 4-byte stores and the 16 B/lane streaming reads the guide's x2 FETCH correction was calibrated on.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/traffic_probe.hip -o tools/probe/traffic_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/probe/traffic_probe   (and WRITE_SIZE)
// Every kernel moves exactly kBytes (64 MiB) of payload per launch:
//   *_stream: every element of a 64 MiB buffer once;   *_repeat: a 1 MiB window 64 times (L2 resident).
// tools/traffic_calibration.py divides the payload by the counters and writes profiles/r03_traffic_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t kBytes = 64ull << 20;
constexpr size_t kWindow = 1ull << 20;

template <typename T> __device__ __forceinline__ T ld_agent(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ void st_wg(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#define LOAD_KERNEL(NAME, T, LD, SPAN)                                                      \
  __global__ void NAME(const T *buf, unsigned long long *sink) {                           \
    const size_t n = kBytes / sizeof(T), span = (SPAN) / sizeof(T);                         \
    unsigned long long acc = 0;                                                             \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;                   \
         i += (size_t)gridDim.x * blockDim.x)                                               \
      acc += (unsigned long long)LD(buf + (i % span));                                      \
    if (acc == 0x123456789abcdefull) *sink = acc;                                           \
  }
#define STORE_KERNEL(NAME, T, ST, SPAN)                                                     \
  __global__ void NAME(T *buf) {                                                            \
    const size_t n = kBytes / sizeof(T), span = (SPAN) / sizeof(T);                         \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;                   \
         i += (size_t)gridDim.x * blockDim.x)                                               \
      ST(buf + (i % span), (T)i);                                                           \
  }
__device__ __forceinline__ unsigned ld_plain(const unsigned *p) { return *p; }

LOAD_KERNEL(load4_agent_stream, unsigned, ld_agent<unsigned>, kBytes)
LOAD_KERNEL(load8_agent_stream, unsigned long long, ld_agent<unsigned long long>, kBytes)
LOAD_KERNEL(load4_agent_repeat, unsigned, ld_agent<unsigned>, kWindow)
LOAD_KERNEL(load8_agent_repeat, unsigned long long, ld_agent<unsigned long long>, kWindow)
LOAD_KERNEL(load4_plain_stream, unsigned, ld_plain, kBytes)
LOAD_KERNEL(load4_plain_repeat, unsigned, ld_plain, kWindow)
STORE_KERNEL(store4_agent_stream, unsigned, st_agent<unsigned>, kBytes)
STORE_KERNEL(store8_agent_stream, unsigned long long, st_agent<unsigned long long>, kBytes)
STORE_KERNEL(store4_agent_repeat, unsigned, st_agent<unsigned>, kWindow)
STORE_KERNEL(store8_agent_repeat, unsigned long long, st_agent<unsigned long long>, kWindow)
STORE_KERNEL(store4_wg_stream, unsigned, st_wg<unsigned>, kBytes)
STORE_KERNEL(store4_wg_repeat, unsigned, st_wg<unsigned>, kWindow)

__global__ void load16_plain_stream(const uint4 *buf, unsigned long long *sink) {
  const size_t n = kBytes / 16;
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = buf[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}
// scattered 4-byte agent-scope stores, one per 64-byte line (what a bid / flag / price update looks like):
// kBytes / 16 stores touching kBytes / 16 distinct lines of a 64 MiB buffer... payload = 4 bytes per store
__global__ void store4_agent_scattered(unsigned *buf) {
  const size_t n = kBytes / 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    st_agent<unsigned>(buf + i * 16, (unsigned)i);
}
__global__ void load4_agent_scattered(const unsigned *buf, unsigned long long *sink) {
  const size_t n = kBytes / 64;
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += ld_agent<unsigned>(buf + i * 16);
  if (acc == 0x123456789abcdefull) *sink = acc;
}

int main() {
  void *buf = nullptr, *sink = nullptr;
  if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
  hipMemset(buf, 1, kBytes);
  hipDeviceSynchronize();
  const dim3 grid(2048), block(256);
  for (int rep = 0; rep < 2; ++rep) {
    load16_plain_stream<<<grid, block>>>((const uint4 *)buf, (unsigned long long *)sink);
    load4_plain_stream<<<grid, block>>>((const unsigned *)buf, (unsigned long long *)sink);
    load4_plain_repeat<<<grid, block>>>((const unsigned *)buf, (unsigned long long *)sink);
    load4_agent_stream<<<grid, block>>>((const unsigned *)buf, (unsigned long long *)sink);
    load8_agent_stream<<<grid, block>>>((const unsigned long long *)buf, (unsigned long long *)sink);
    load4_agent_repeat<<<grid, block>>>((const unsigned *)buf, (unsigned long long *)sink);
    load8_agent_repeat<<<grid, block>>>((const unsigned long long *)buf, (unsigned long long *)sink);
    load4_agent_scattered<<<grid, block>>>((const unsigned *)buf, (unsigned long long *)sink);
    store4_wg_stream<<<grid, block>>>((unsigned *)buf);
    store4_wg_repeat<<<grid, block>>>((unsigned *)buf);
    store4_agent_stream<<<grid, block>>>((unsigned *)buf);
    store8_agent_stream<<<grid, block>>>((unsigned long long *)buf);
    store4_agent_repeat<<<grid, block>>>((unsigned *)buf);
    store8_agent_repeat<<<grid, block>>>((unsigned long long *)buf);
    store4_agent_scattered<<<grid, block>>>((unsigned *)buf);
    hipDeviceSynchronize();
  }
  printf("traffic probe done: %zu payload bytes per kernel (scattered: %zu)\n", kBytes, kBytes / 16);
  return 0;
}
