// Library identity + error reporting for libnesie_hip.so.
#include "common.h"
#include <stdarg.h>

namespace nesie {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace nesie

extern "C" int nesie_abi_version(void) { return 1; }
extern "C" const char *nesie_last_error(void) { return nesie::g_err; }
