// pw_fwd_kernel<64, 4, 2, 64, *, *>: K <= 256, 4 x 2 waves, 64-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(64, 4, 2, 64)
