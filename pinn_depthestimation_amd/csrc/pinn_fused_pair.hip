// pinn_fused_pair.hip — instantiations of the two-waves-per-SIMD fused kernel (fused_pair_kernel.h),
// hidden width padded to 64, even jets (1+k in {2,4})
#include <type_traits>
#include "fused_pair_kernel.h"

namespace pinn {

template <int K1, bool GRAD, int ACT>
static int launch_pair_act(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  auto kern = k_fused_pair<64, K1, GRAD, ACT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return PINN_ERR_LAUNCH; }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(PR_THREADS), lds, s, P);
  return check_launch("fused pair kernel");
}

template <int K1, bool GRAD>
static int launch_pair(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  return P.act == PINN_ACT_TANH ? launch_pair_act<K1, GRAD, PINN_ACT_TANH>(P, grid, lds, s)
                                : launch_pair_act<K1, GRAD, PINN_ACT_LEAKY_RELU>(P, grid, lds, s);
}

int launch_fused_pair(int K1, bool grad, const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  if (K1 == 4) return grad ? launch_pair<4, true>(P, grid, lds, s) : launch_pair<4, false>(P, grid, lds, s);
  if (K1 == 2) return grad ? launch_pair<2, true>(P, grid, lds, s) : launch_pair<2, false>(P, grid, lds, s);
  set_error("fused pair engine: no kernel for K1=%d", K1);
  return PINN_ERR_UNSUPPORTED;
}

}  // namespace pinn
