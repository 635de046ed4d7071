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
static int g_distance_form = 0;
int distance_form() { return g_distance_form; }
static int g_cu_count = 256;
int cu_count() { return g_cu_count; }
}  // namespace nesie

extern "C" int nesie_abi_version(void) { return 1; }
extern "C" int nesie_set_distance_form(int form) {
  NESIE_REQUIRE(form >= 0 && form <= 2, "set_distance_form");
  nesie::g_distance_form = form;
  return NESIE_OK;
}
extern "C" int nesie_get_distance_form(void) { return nesie::g_distance_form; }
extern "C" int nesie_set_cu_count(int n) {
  NESIE_REQUIRE(n >= 8 && n <= 256 && n % 8 == 0, "set_cu_count");
  nesie::g_cu_count = n;
  return NESIE_OK;
}
extern "C" int nesie_get_cu_count(void) { return nesie::g_cu_count; }
extern "C" const char *nesie_last_error(void) { return nesie::g_err; }
