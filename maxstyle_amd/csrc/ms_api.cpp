// Library-level entry points: version and the thread-local error string.
#include <cstdarg>
#include <cstdio>
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ms

extern "C" int ms_version(void) { return 100; }
extern "C" const char* ms_last_error(void) { return ms::g_err; }
