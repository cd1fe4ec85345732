// simt.h -- SIMT-on-CPU emulation layer used ONLY to generate golden vectors.
//
// Own code.  It lets the text of the reference's CUDA kernels (read at run time
// from /root/reference by tests/golden/gen_emulated.py, never copied into this
// repository) execute on the host: one OS thread per CUDA thread, one block at a
// time, std::barrier as __syncthreads().  This is an EMULATION of the kernels,
// not a build of the reference: scheduling-dependent behaviour (races the
// kernels formally contain) is resolved by whatever the host threads do, so the
// generator screens fixtures for such events (see gen_emulated.py).
#pragma once
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 threadIdx;
static dim3 blockIdx, blockDim, gridDim;
static std::barrier<> *g_barrier = nullptr;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static
static inline void __syncthreads() { g_barrier->arrive_and_wait(); }

static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicCAS(int *p, int cmp, int v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
static inline int atomicExch(int *p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline float atomicAdd(float *p, float v) {
  int *ip = reinterpret_cast<int *>(p);
  int old = __atomic_load_n(ip, __ATOMIC_SEQ_CST);
  for (;;) {
    float f;
    std::memcpy(&f, &old, 4);
    f += v;
    int nv;
    std::memcpy(&nv, &f, 4);
    if (__atomic_compare_exchange_n(ip, &old, nv, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
      std::memcpy(&f, &old, 4);
      return f;
    }
  }
}
static inline int __float_as_int(float f) {
  int i;
  std::memcpy(&i, &f, 4);
  return i;
}
static inline float __int_as_float(int i) {
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}
using std::max;
using std::min;

template <typename K, typename... Args>
static void simt_launch(K kernel, dim3 grid, dim3 block, Args... args) {
  gridDim = grid;
  blockDim = block;
  const unsigned nthreads = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        std::barrier<> bar(nthreads);
        g_barrier = &bar;
        std::vector<std::thread> ts;
        ts.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; ++t)
          ts.emplace_back([=]() {
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            kernel(args...);
          });
        for (auto &th : ts) th.join();
      }
}

// tiny binary IO: file = sequence of raw arrays in a fixed order
static inline void read_all(const char *path, std::vector<char> &buf) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(n);
  if (fread(buf.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
}
