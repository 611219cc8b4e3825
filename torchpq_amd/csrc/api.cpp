// Version / error plumbing of libtorchpq_amd.so.
#include <stdarg.h>

#include "common.h"

namespace tpq {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tpq

extern "C" int tpq_version(void) { return TPQ_VERSION; }
extern "C" const char* tpq_last_error(void) { return tpq::g_err; }
