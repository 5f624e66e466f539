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

// Per-stream scratch of the two-level amax / loss-sum reductions (common.h: AmaxScratch).  A pool of 64 zero-filled areas;
// a stream keeps the area it is given.  The pool is allocated by mt_init() (the host side calls it when it loads the
// library on a GPU box: allocation and memset are not capturable, and a lazy allocation inside a launch path could break
// ANOTHER stream's global-mode capture -- ADVICE round 3); a launch that finds no pool still tries once, outside captures.
// nullptr (callers fall back to raise_amax) while there is no pool or when it is used up.
namespace {
constexpr int POOL = 64;
std::mutex g_mu;
std::unordered_map<hipStream_t, AmaxScratch*> g_map;
AmaxScratch* g_pool = nullptr;
int g_used = 0, g_pool_dev = -1;
bool g_failed = false;

bool pool_alloc_locked(int dev) {
  if (g_pool) return true;
  if (g_failed) return false;
  if (hipMalloc(&g_pool, sizeof(AmaxScratch) * POOL) != hipSuccess || hipMemset(g_pool, 0, sizeof(AmaxScratch) * POOL) != hipSuccess) {
    (void)hipGetLastError();
    g_pool = nullptr;
    g_failed = true;
    return false;
  }
  g_pool_dev = dev;
  return true;
}
}  // namespace

extern "C" int mt_init(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    mt_set_error("mt_init: no HIP device");
    return MT_ERR_LAUNCH;
  }
  if (!pool_alloc_locked(dev)) {
    mt_set_error("mt_init: could not allocate the reduction scratch (%zu bytes)", sizeof(AmaxScratch) * POOL);
    return MT_ERR_WORKSPACE;
  }
  return MT_OK;
}

// The scratch is persistent cross-launch state: every launch leaves it zero-filled, but a launch that was aborted or
// killed half way does not.  This re-zeroes the whole pool on ``stream`` (not inside a capture).
extern "C" int mt_amax_scratch_reset(void* stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_pool) return MT_OK;
  if (hipMemsetAsync(g_pool, 0, sizeof(AmaxScratch) * POOL, static_cast<hipStream_t>(stream)) != hipSuccess) {
    mt_set_error("mt_amax_scratch_reset: %s", hipGetErrorString(hipGetLastError()));
    return MT_ERR_LAUNCH;
  }
  return MT_OK;
}

AmaxScratch* mt_amax_scratch(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_mu);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (g_pool && dev != g_pool_dev) return nullptr;      // (one process per GPU: a second device of the same process goes without)
  auto it = g_map.find(s);
  if (it != g_map.end()) return it->second;
  if (!g_pool && !g_failed) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;                       // (not remembered: the stream gets an area once the pool exists)
    }
    pool_alloc_locked(dev);
  }
  AmaxScratch* p = (g_pool && g_used < POOL) ? g_pool + g_used++ : nullptr;
  g_map[s] = p;
  return p;
}
