// 128x128-tile grouped GEMM kernels (instantiation unit of gemm_kernels.h)
#include "gemm_kernels.h"
namespace phg { namespace reg {
int launch_grouped_128(const GroupParams& g, int total, int max_blocks, int ta, int tb, int pf, int conv, hipStream_t s) {
  return launch_grouped_any<128>(g, total, max_blocks, ta, tb, pf, conv, s);
}
} }  // namespace phg::reg
