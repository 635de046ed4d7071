// pw_fwd_kernel<32, 4, 2, 128, *, *>: K <= 128, 4 x 2 waves, 128-position tiles (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(32, 4, 2, 128)
