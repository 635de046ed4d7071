// pw_fwd_kernel<16, 4, 2, 256, *, *>: K <= 64, 4 x 2 waves, 256-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(16, 4, 2, 256)
