// Error plumbing and version of libsnws.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void snws_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int snws_version(void) { return 100; }
extern "C" const char* snws_last_error(void) { return g_err; }
