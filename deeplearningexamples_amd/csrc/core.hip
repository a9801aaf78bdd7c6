// C-ABI plumbing shared by every kernel file: thread-local error string, version, device probe.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void dle_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dle_last_error(void) { return g_err; }

extern "C" int dle_abi_version(void) { return 1; }

// 0 when device `dev` is a gfx950 (MI355X-class) part; fills name[] with gcnArchName.
extern "C" int dle_device_check(int dev, char* name, int name_len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    dle_set_error("hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
    return (int)e;
  }
  if (name && name_len > 0) snprintf(name, name_len, "%s", p.gcnArchName);
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 0 : 1;
}
