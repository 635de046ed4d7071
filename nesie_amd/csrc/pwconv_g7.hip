// pw_fwd_kernel<64, 8, 1, 64, *, *>: K <= 256, 8 x 1 waves, 64-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(64, 8, 1, 64)
