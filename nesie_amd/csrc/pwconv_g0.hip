// pw_fwd_kernel<4, 1, 4, 1, 1, *, *>: K sub-tile / 16, sub-tiles along K, row waves, column waves, 16-row sets
// per wave (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(4, 1, 4, 1, 1)
