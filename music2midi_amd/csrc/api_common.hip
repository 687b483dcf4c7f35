// Error reporting and small ABI entry points shared by every translation unit.
#include "common.h"

#include <stdarg.h>

namespace m2m {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace m2m

extern "C" int m2m_abi_version(void) { return M2M_ABI_VERSION; }
extern "C" const char* m2m_last_error(void) { return m2m::g_err; }
extern "C" int m2m_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
