// Error plumbing and library identity for libasis_hip.so.
#include <stdarg.h>

#include "asis_common.h"

static thread_local char g_err[512] = "";

extern "C" void asis_set_error_(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* asis_last_error(void) { return g_err; }
extern "C" int asis_version(void) { return 100; }

extern "C" int asis_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    asis_set_error_("hipGetDeviceCount: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return ASIS_ELAUNCH;
  }
  return n;
}
