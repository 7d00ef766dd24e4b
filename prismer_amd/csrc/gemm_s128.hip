// 128x128 / 128x64-tile single-launch GEMM kernels (instantiation unit of gemm_kernels.h)
#include "gemm_kernels.h"
namespace phg { namespace reg {
int launch_single_128(const GemmParams& p, int bn, int ta, int tb, int splits, hipStream_t s) {
  return bn == 64 ? dispatch_layout<128, 64>(p, ta, tb, splits, s) : dispatch_layout<128, 128>(p, ta, tb, splits, s);
}
} }  // namespace phg::reg
