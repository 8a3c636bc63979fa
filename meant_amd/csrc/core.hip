// Library-wide plumbing: thread-local error message, version, device properties.
#include "common.h"
#include <stdarg.h>
#include <mutex>

static thread_local char g_err[512] = "";

void meant_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* meant_last_error(void) { return g_err; }

extern "C" int meant_version(void) { return 100; /* 0.1.0 */ }

extern "C" int meant_num_cus(void) {
  static std::once_flag once;
  static int cus[64];
  std::call_once(once, [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    for (int i = 0; i < 64; ++i) cus[i] = 0;
    for (int i = 0; i < n && i < 64; ++i) {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, i) == hipSuccess) cus[i] = p.multiProcessorCount;
    }
  });
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  return cus[dev];
}
