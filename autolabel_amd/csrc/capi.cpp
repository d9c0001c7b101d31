// Error reporting and ABI version for libautolabel_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/autolabel_hip.h"

static thread_local char g_err[512] = "";

void aln_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* aln_last_error(void) { return g_err; }
extern "C" int aln_abi_version(void) { return ALN_ABI_VERSION; }
