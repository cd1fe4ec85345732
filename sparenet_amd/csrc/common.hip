// common.hip -- error plumbing and ABI version of libsparenet_hip.so.
#include "common.hpp"

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace sn {

char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
struct Span {
  hipEvent_t a, b;
};
std::atomic<bool> g_prof{false};
std::mutex g_mu;
std::map<std::string, std::vector<Span>> g_spans;
std::map<std::string, hipEvent_t> g_open;
}  // namespace

bool prof_enabled() { return g_prof.load(std::memory_order_relaxed); }

void prof_begin(const char *name, hipStream_t s) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_open[name] = e;
}

void prof_end(const char *name, hipStream_t s) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_open.find(name);
  if (it == g_open.end()) {
    (void)hipEventDestroy(e);
    return;
  }
  g_spans[name].push_back({it->second, e});
  g_open.erase(it);
}

namespace {
struct Sticky {
  unsigned *host = nullptr, *dev = nullptr;
  bool tried = false;
};
std::mutex g_sticky_mu;
Sticky g_sticky[64];
}  // namespace

unsigned *sticky_device_word(int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_sticky_mu);
  Sticky &st = g_sticky[dev];
  if (!st.tried) {
    st.tried = true;
    void *h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess) {
      *static_cast<volatile unsigned *>(h) = 0u;
      void *d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
        st.host = static_cast<unsigned *>(h);
        st.dev = static_cast<unsigned *>(d);
      }
    }
    (void)hipGetLastError();
  }
  return st.dev;
}

int check_sticky(int dev, const char *what) {
  if (dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_sticky_mu);
  unsigned *w = g_sticky[dev].host;
  if (w && *reinterpret_cast<volatile unsigned *>(w) != 0u) {
    *reinterpret_cast<volatile unsigned *>(w) = 0u;
    return fail(SN_ETIMEDOUT,
                "%s: a bounded wait inside an EARLIER multi-workgroup launch on this device timed out (persistent EMD "
                "auction or density sampler; the device is shared, or a debugger holds a compute unit); that call's "
                "outputs were filled with NaN distances / -1 assignments (EMD) or whole index rows of -1 (sampler; "
                "sn_gather_forward turns those into NaN features)", what);
  }
  return 0;
}

namespace {
struct Chain {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  bool recorded = false;
};
Chain g_chain[64];
}  // namespace

PersistentLaunch::PersistentLaunch(int dev, hipStream_t stream) : dev_(dev), stream_(stream), chained_(false) {
  if (dev_ < 0 || dev_ >= 64) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream_, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return;
  }
  Chain &c = g_chain[dev_];
  c.mu.lock();
  chained_ = true;
  if (!c.ev && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) {
    c.ev = nullptr;
    (void)hipGetLastError();
  }
  if (c.ev && c.recorded) (void)hipStreamWaitEvent(stream_, c.ev, 0);
}

PersistentLaunch::~PersistentLaunch() {
  if (!chained_) return;
  Chain &c = g_chain[dev_];
  if (c.ev && hipEventRecord(c.ev, stream_) == hipSuccess) c.recorded = true;
  c.mu.unlock();
}

void clear_sticky(int dev) {
  if (dev < 0 || dev >= 64) return;
  std::lock_guard<std::mutex> lk(g_sticky_mu);
  if (g_sticky[dev].host) *reinterpret_cast<volatile unsigned *>(g_sticky[dev].host) = 0u;
}

}  // namespace sn

extern "C" void sn_prof_enable(int on) { sn::g_prof.store(on != 0); }

// Sum of the recorded launch durations of `name` since the last reset; waits for the
// recorded events.  Returns the number of launches (0 if none), total_ms may be NULL.
extern "C" long long sn_prof_read(const char *name, double *total_ms) {
  std::lock_guard<std::mutex> lk(sn::g_mu);
  auto it = sn::g_spans.find(name ? name : "");
  if (it == sn::g_spans.end()) {
    if (total_ms) *total_ms = 0.0;
    return 0;
  }
  double tot = 0.0;
  for (auto &sp : it->second) {
    (void)hipEventSynchronize(sp.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) tot += ms;
  }
  if (total_ms) *total_ms = tot;
  return (long long)it->second.size();
}

extern "C" void sn_prof_reset(void) {
  std::lock_guard<std::mutex> lk(sn::g_mu);
  for (auto &kv : sn::g_spans)
    for (auto &sp : kv.second) {
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
  sn::g_spans.clear();
  for (auto &kv : sn::g_open) (void)hipEventDestroy(kv.second);
  sn::g_open.clear();
}

#ifndef SN_BUILD_ID
#define SN_BUILD_ID "unknown"
#endif
// 0, or SN_ETIMEDOUT when a bounded wait of an earlier multi-workgroup launch on the current device gave up (clears the
// word).  Reads one word of pinned host memory: no synchronisation.  Meaningful AFTER the work in question has
// finished -- call it where the host already waits for the GPU (reading a loss, an optimiser step's `.item()`).
extern "C" int sn_device_status(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return sn::check_sticky(dev, "sn_device_status");
}

extern "C" int sn_abi_version(void) { return SN_ABI_VERSION; }
extern "C" const char *sn_build_id(void) { return SN_BUILD_ID; }
extern "C" const char *sn_last_error(void) { return sn::last_error_buf(); }
