// pw_fwd_kernel<32, 4, 2, 64, *, *>: half-size operand tiles, two workgroups per CU (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(32, 4, 2, 64)
