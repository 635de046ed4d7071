// pw_fwd_kernel<33, 4, 2, 128, *, *>: K <= 132, 4 x 2 waves, 128-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(33, 4, 2, 128)
