// Error plumbing + version for libprismer_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local std::string g_last_error;

void ph_set_error(const std::string& msg) { g_last_error = msg; }

int ph_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

extern "C" int ph_version(void) { return PH_VERSION; }
extern "C" const char* ph_last_error(void) { return g_last_error.c_str(); }
