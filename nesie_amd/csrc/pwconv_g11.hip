// pw_fwd_kernel<33, 8, 1, 64, *, *>: half-size operand tiles, two workgroups per CU (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(33, 8, 1, 64)
