// pw_fwd_kernel<9, 1, 8, 1, 1, *, *>: K sub-tile / 16, sub-tiles along K, row waves, column waves, 16-row sets
// per wave (pwconv_fwd.h)
#include "pwconv_fwd.h"
PW_GEOM_DEF(9, 1, 8, 1, 1)
