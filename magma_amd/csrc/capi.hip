// capi.hip -- library-level entry points of libmagma_hip.so (version, error string).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mg_version(void) { return "magma_hip 0.1 (gfx950)"; }
extern "C" const char* mg_last_error(void) { return g_err; }
