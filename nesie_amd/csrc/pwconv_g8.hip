// pw_fwd_kernel<65, 4, 2, 64, *, *>: K <= 260, 4 x 2 waves, 64-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(65, 4, 2, 64)
