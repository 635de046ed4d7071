// Library identity + error reporting for libnesie_hip.so.
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>

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

// ---- reversed tile order for big operands (round 5) ------------------------------------------------
// A tensor larger than about half the 256 MB Infinity Cache is only partly resident when the next
// launch reads it: what stays is what was written LAST.  Every producer of the step writes its
// output from the first position to the last, so a persistent launch whose operand is that big walks
// its tiles from the LAST to the first: it starts on cached data and reaches the evicted part at its
// end (measured: the step 12.49 -> 12.36 ms, same-box A/B, NESIE_PW_REV_MB=0 vs 100).  Stateless:
// the direction is a function of the operand's size alone, so a launch sums the same tiles per
// workgroup in every run.  (A "serpentine" variant that tracked the direction each tensor was
// written in -- so that a reversed producer's consumer runs forwards -- measured the same -0.13 ms
// and made the association of the partial sums depend on allocator history: dropped.)
static const int g_rev_mb = [] { const char *e = getenv("NESIE_PW_REV_MB"); return e ? atoi(e) : 100; }();
int walk_dir(long long operand_bytes) {
  return g_rev_mb > 0 && operand_bytes >= (long long)g_rev_mb * 1000000 ? 1 : 0;
}
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
