// Error plumbing of the C ABI: thread-local message, never throws.
#include <cstdarg>
#include <cstdio>

#include "../../include/danhip.h"

static thread_local char g_err[512] = "";

void danhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* danhip_last_error(void) { return g_err; }
extern "C" int danhip_version(void) { return 1; }
extern "C" int danhip_act_dtype(void) {
#ifdef DANHIP_FP16
  return DANHIP_F16;
#else
  return DANHIP_BF16;
#endif
}
