// Error plumbing + version for libbiapy_amd.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/biapy_amd.h"

static thread_local char g_err[512] = "";

void bpx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* bpx_last_error(void) { return g_err; }
extern "C" int bpx_version(void) { return 100; }
