// common.hpp -- shared host/device helpers for libsparenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/sparenet_hip.h"

namespace sn {

constexpr int kWave = 64;  // CDNA wavefront width

// thread-local last-error text, exposed through sn_last_error()
char *last_error_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// returns 0 or records + returns the pending launch error
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

#define SN_REQUIRE(cond, ...)                        \
  do {                                               \
    if (!(cond)) return sn::fail(SN_EINVAL, __VA_ARGS__); \
  } while (0)

#define SN_HIP(call)                                                       \
  do {                                                                     \
    hipError_t e__ = (call);                                               \
    if (e__ != hipSuccess)                                                 \
      return sn::fail((int)e__, "%s: %s", #call, hipGetErrorString(e__));  \
  } while (0)

// ---- optional per-kernel timing (off by default; used by bench.py's roofline leg).
// When enabled, the heavy kernels are bracketed by hipEventRecord on their own stream.
bool prof_enabled();
void prof_begin(const char *name, hipStream_t s);
void prof_end(const char *name, hipStream_t s);

#define SN_TIMED(name, stream, launch_expr)                \
  do {                                                     \
    if (sn::prof_enabled()) sn::prof_begin(name, stream);  \
    launch_expr;                                           \
    if (sn::prof_enabled()) sn::prof_end(name, stream);    \
  } while (0)

// Per-device sticky error word (pinned host memory, device-visible): a kernel whose workgroups wait for each other
// (the persistent EMD auction, the multi-workgroup density sampler) sets it when a bounded wait gives up; the next
// call of such an op on the device returns SN_ETIMEDOUT without a host synchronisation.
unsigned *sticky_device_word(int dev);            // nullptr if the word could not be allocated
int check_sticky(int dev, const char *what);      // 0, or SN_ETIMEDOUT (clears the word, fills sn_last_error)
void clear_sticky(int dev);

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace sn
