// Library plumbing: version string and per-thread error message.
#include <stdarg.h>
#include "iso_common.h"

static thread_local char g_iso_err[512] = "";

void iso_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_iso_err, sizeof(g_iso_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* iso_version(void) { return "isopoints-hip 0.1 (gfx950)"; }
extern "C" const char* iso_last_error(void) { return g_iso_err; }
