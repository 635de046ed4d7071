// pw_fwd_kernel<64, 8, 1, 32, *, *>: half-size operand tiles, two workgroups per CU (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(64, 8, 1, 32)
