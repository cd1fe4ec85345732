// common.hip -- error plumbing and ABI version of libsparenet_hip.so.
#include "common.hpp"

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

}  // namespace sn

extern "C" int sn_abi_version(void) { return SN_ABI_VERSION; }
extern "C" const char *sn_last_error(void) { return sn::last_error_buf(); }
