// capi.hip -- library-level entry points of libmagma_hip.so (version, error string).
#include "common.h"
#include <string.h>
#include <mutex>
#include <set>
#include <utility>

static thread_local char g_err[512] = "";

void mg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mg_version(void) { return "magma_hip 0.2 (gfx950)"; }
extern "C" int32_t mg_abi_version(void) { return MG_ABI_VERSION; }
extern "C" const char* mg_last_error(void) { return g_err; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device).  The launchers used to keep a process-wide
// `static bool` each: unsynchronised, and wrong on a second GPU (the attribute belongs to the device's code object).
int mg_allow_dynamic_lds(const void* fn, int bytes, const char* who) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) MG_FAIL(MG_ERR_HIP, "%s: hipGetDevice: %s", who, hipGetErrorString(e));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({fn, dev})) return MG_OK;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) MG_FAIL(MG_ERR_HIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
  done.insert({fn, dev});
  return MG_OK;
}
