// pw_fwd_kernel<6, 2, 4, 1, 1, sparse>: the input-gradient launch of SA1's pooled tail (128 built
// rows + 64 loaded rows -> 64 output rows; pool_tail.hip)
#include "pwconv_fwd.h"
namespace nesie {
int pw_launch_sparse_6_2_4_1_1(const PwFwd &a, int grid, size_t lds, hipStream_t s) {
  return pw_launch_sparse<6, 2, 4, 1, 1, PW_SPARSE128>(a, grid, lds, s);
}
}  // namespace nesie
