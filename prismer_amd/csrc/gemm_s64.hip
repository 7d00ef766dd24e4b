// 64x64-tile single-launch GEMM kernels and the intra-block k-split form (instantiation unit of gemm_kernels.h)
#include "gemm_kernels.h"
namespace phg { namespace reg {
int launch_single_64(const GemmParams& p, int ta, int tb, int splits, hipStream_t s) { return dispatch_layout<64, 64>(p, ta, tb, splits, s); }
int launch_ks2(const GemmParams& p, int tb, hipStream_t s) { return tb ? phg::launch_ks2<false, true>(p, s) : phg::launch_ks2<false, false>(p, s); }
} }  // namespace phg::reg
#ifdef PH_TIMELINE
extern "C" int ph_tl_fetch_gemm(unsigned long long* host, int n, int reset) {
  hipDeviceSynchronize();
  if (host && n > 0) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * (size_t)n);
  if (reset) { void* d = nullptr; hipGetSymbolAddress(&d, HIP_SYMBOL(g_tl)); hipMemset(d, 0, sizeof(g_tl)); }
  return 0;
}
#endif
