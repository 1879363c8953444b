// Error text + version for libbuglab_hip.
#include <stdarg.h>

#include "bl_common.h"

static thread_local char g_err[512] = "";

void bl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* bl_last_error(void) { return g_err; }
extern "C" int bl_version(void) { return 1; }
