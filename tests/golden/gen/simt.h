// simt.h -- SIMT-on-CPU emulation layer used ONLY to generate golden vectors.
//
// Own code.  It lets the text of the reference's CUDA kernels (read at run time
// from /root/reference by tests/golden/gen_emulated.py, never copied into this
// repository) execute on the host: one OS thread per CUDA thread, one block at a
// time, std::barrier as __syncthreads().  This is an EMULATION of the kernels,
// not a build of the reference: scheduling-dependent behaviour (races the
// kernels formally contain) is resolved by whatever the host threads do, so the
// generator screens fixtures for such events (see gen_emulated.py).
//
// Schedule knob: SIMT_SCHED_SEED=<n> (n > 0) in the environment starts the threads of every block in a
// random order and injects sched_yield() at barriers and atomics with probability 1/8 per thread and event
// (a per-thread LCG seeded from n).  gen_emulated.py runs every fixture under several seeds and only calls
// a fixture schedule invariant if all runs agree byte for byte.
#pragma once
#include <sched.h>

#include <random>
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
static const unsigned g_sched_seed = getenv("SIMT_SCHED_SEED") ? (unsigned)atoi(getenv("SIMT_SCHED_SEED")) : 0u;
static thread_local uint64_t t_sched_rng = 0;
static inline void simt_perturb() {
  if (!g_sched_seed) return;
  t_sched_rng = t_sched_rng * 6364136223846793005ULL + 1442695040888963407ULL;
  if (((t_sched_rng >> 33) & 7u) == 0u) sched_yield();
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static
static inline void __syncthreads() {
  simt_perturb();
  g_barrier->arrive_and_wait();
  simt_perturb();
}

static inline int atomicAdd(int *p, int v) {
  simt_perturb();
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline int atomicCAS(int *p, int cmp, int v) {
  simt_perturb();
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
static inline int atomicExch(int *p, int v) {
  simt_perturb();
  return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST);
}
static inline float atomicAdd(float *p, float v) {
  simt_perturb();
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
        std::vector<unsigned> order(nthreads);
        for (unsigned t = 0; t < nthreads; ++t) order[t] = t;
        if (g_sched_seed) {
          std::mt19937 rng(g_sched_seed * 7919u + bx + 131u * by + 17161u * bz);
          std::shuffle(order.begin(), order.end(), rng);
        }
        for (unsigned k = 0; k < nthreads; ++k) {
          const unsigned t = order[k];
          ts.emplace_back([=]() {
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            t_sched_rng = ((uint64_t)g_sched_seed << 32) ^ (0x9E3779B97F4A7C15ULL * (t + 1));
            kernel(args...);
          });
        }
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
