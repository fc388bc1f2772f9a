// C-ABI plumbing: per-thread error string, version, kernel timing hooks.
#include <stdarg.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "egt_common.h"

static thread_local char g_err[512] = "";

void egt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* egt_last_error_string(void) { return g_err; }
extern "C" int egt_abi_version(void) { return EGT_ABI_VERSION; }
