// pw_fwd_kernel<32, 8, 1, 128, *, *>: K <= 128, 8 x 1 waves, 128-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(32, 8, 1, 128)
