// Library-level entry points: ABI version and the per-thread error string.
#include <stdarg.h>
#include "common.h"

namespace cl4 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cl4

extern "C" int cl4_abi_version(void) { return 1; }
extern "C" const char* cl4_last_error(void) { return cl4::g_err; }
