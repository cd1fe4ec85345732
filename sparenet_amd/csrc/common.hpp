// common.hpp -- shared host/device helpers for libsparenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

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

// Launches whose workgroups WAIT for each other (the persistent EMD auction, the sampler's dense-regime teams) must
// not overlap each other on a device: each may hold compute units with members of a not-yet-complete team, and two
// such launches on two streams could hold all of them between them -- neither team ever completes, both give up
// after their spin limit.  (Ordinary kernels on other streams are harmless: they finish and free their CUs.)
// PersistentLaunch orders them on the GPU without a host synchronisation: the constructor makes `stream` wait for
// the event of the previous such launch of this process on the device, the destructor records the event behind the
// new launch; a host mutex is held in between, so launches from several host threads chain one after the other.
// Skipped while `stream` is being captured into a graph (an event recorded outside the capture cannot be waited
// for inside it); launches of OTHER processes sharing the GPU are beyond its reach -- for those the bounded waits
// and the sticky error word remain.
class PersistentLaunch {
 public:
  PersistentLaunch(int dev, hipStream_t stream);
  ~PersistentLaunch();
  PersistentLaunch(const PersistentLaunch &) = delete;
  PersistentLaunch &operator=(const PersistentLaunch &) = delete;

 private:
  int dev_;
  hipStream_t stream_;
  bool chained_;
};

// true while `stream` is being captured into a HIP graph
inline bool capturing(hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool yes = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  if (!yes) (void)hipGetLastError();
  return yes;
}

// SN_ALLOW_CAPTURE=1 lifts the refusals below -- for graphs built with the HIP graph API, where every call of the
// step replays bit-identically (tools/probe/graph_emd.hip, tests/test_graph_capture.py).  NOT for
// torch.cuda.CUDAGraph: under PyTorch 2.10 / ROCm 7.2 the auction's second replay hangs and Chamfer forward +
// backward through autograd faults; a one-time warning on stderr says so.
inline bool capture_allowed() {
  static const bool on = [] {
    const char *e = getenv("SN_ALLOW_CAPTURE");
    const bool v = e && e[0] == '1';
    if (v)
      fprintf(stderr, "sparenet_hip: SN_ALLOW_CAPTURE=1 -- graph capture of the auction / sorted Chamfer is allowed. Supported "
                      "for HIP-level graphs (hipGraphLaunch) only: replays under torch.cuda.CUDAGraph hang or fault.\n");
    return v;
  }();
  return on;
}

// Refused by default because torch is how a user would capture: through the raw HIP graph API these ops replay
// correctly (section 1 of DESIGN.md), under torch.cuda.CUDAGraph (PyTorch 2.10 on ROCm 7.2 / gfx950) the persistent
// auction's second replay runs into its barrier time-outs and the Chamfer kernels' replay through autograd dies with a
// memory access fault, although the same launches are clean in eager mode with every input at the end of its
// allocation (tools/oob_probe.py).  They refuse instead of producing a graph that misbehaves later.
#define SN_REFUSE_CAPTURE(stream, what)                                                                     \
  SN_REQUIRE(!sn::capturing(stream) || sn::capture_allowed(), what ": the stream is being captured into a HIP graph; this op does not " \
                                           "replay correctly from a graph (see common.hpp) -- launch it eagerly")

// Tuning / test knobs from the environment are read ONCE per process, at their first use -- unless
// SN_KNOBS_PER_CALL=1 (tests/conftest.py sets it before the library is loaded: the tests switch knobs inside one
// process).  SN_KNOB("NAME") = the value (a private copy) or nullptr.
inline bool knobs_per_call() {
  static const bool v = [] { const char *e = getenv("SN_KNOBS_PER_CALL"); return e && e[0] == '1'; }();
  return v;
}
#define SN_KNOB(name)                                                                   \
  ([]() -> const char * {                                                               \
    if (sn::knobs_per_call()) return getenv(name);                                      \
    static const std::string v = [] {                                                   \
      const char *e = getenv(name);                                                     \
      return e ? std::string(e) : std::string("\x01");                                  \
    }();                                                                                \
    return v[0] == '\x01' ? nullptr : v.c_str();                                        \
  }())

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace sn
