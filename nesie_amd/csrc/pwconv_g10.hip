// pw_fwd_kernel<32, 8, 1, 64, *, *>: K <= 128, 8 x 1 waves, 64-position tiles (32 KB operand
// tiles: two workgroups share a CU and fill each other's barrier / epilogue phases)
#include "pwconv_fwd.h"
PW_GEOM_DEF(32, 8, 1, 64)
