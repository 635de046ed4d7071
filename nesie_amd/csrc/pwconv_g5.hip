// pw_fwd_kernel<33, 8, 1, 128, *, *>: K <= 132, 8 x 1 waves, 128-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(33, 8, 1, 128)
