// Error reporting and version of the mt355 C ABI.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "mt355.h"
#include "common.h"
#include <mutex>
#include <unordered_map>

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

extern "C" int mt_version(void) { return 2; }
extern "C" const char* mt_last_error(void) { return g_err; }

// Per-stream scratch of the two-level amax reduction (common.h: AmaxScratch).  A pool of 64 zero-filled areas is
// allocated by the first call that is not inside a graph capture (allocation and memset are not capturable); a stream
// keeps the area it is given.  nullptr (callers fall back to raise_amax) while there is no pool or when it is used up.
AmaxScratch* mt_amax_scratch(hipStream_t s) {
  constexpr int POOL = 64;
  static std::mutex mu;
  static std::unordered_map<hipStream_t, AmaxScratch*> map;
  static AmaxScratch* pool = nullptr;
  static int used = 0, pool_dev = -1;
  static bool failed = false;
  std::lock_guard<std::mutex> lock(mu);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (pool && dev != pool_dev) return nullptr;      // (one process per GPU: a second device of the same process goes without)
  auto it = map.find(s);
  if (it != map.end()) return it->second;
  if (!pool && !failed) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;                       // (not remembered: the stream gets an area once the pool exists)
    }
    if (hipMalloc(&pool, sizeof(AmaxScratch) * POOL) != hipSuccess || hipMemset(pool, 0, sizeof(AmaxScratch) * POOL) != hipSuccess) {
      (void)hipGetLastError();
      pool = nullptr;
      failed = true;
    }
    pool_dev = dev;
  }
  AmaxScratch* p = (pool && used < POOL) ? pool + used++ : nullptr;
  map[s] = p;
  return p;
}
