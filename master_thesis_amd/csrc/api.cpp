// Error reporting and version of the mt355 C ABI.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "mt355.h"

static thread_local char g_err[512] = "";

void mt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mt_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mt_set_error("%s: %s", what, hipGetErrorString(e));
    return MT_ERR_LAUNCH;
  }
  return MT_OK;
}

extern "C" int mt_version(void) { return 1; }
extern "C" const char* mt_last_error(void) { return g_err; }
