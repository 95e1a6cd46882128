// pinn_fused_w64_drop.hip — the fused chain kernel with nn.Dropout(p > 0) in training mode (fused_kernel.h, DROP):
// padded hidden width 64, tanh, gradient passes (the harness's loss + gradient calls; train.py:186 puts the module in
// training mode).  Forward-only and jet calls with dropout stay on the generic engine, whose pinn_jet_backward is the
// other half of the autograd route.  Own translation unit: the p = 0 kernels of pinn_fused_w64.hip are untouched.
#include <type_traits>
#include "fused_kernel.h"

namespace pinn {

template <int K1>
static int launch_drop(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  auto kern = k_fused<64, K1, true, true, PINN_ACT_TANH, EPI_GENERIC, 0, true>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), lds, s, P);
  return check_launch("fused kernel (WP=64, dropout)");
}

int launch_fused_drop64(int K1, const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  switch (K1) {
    case 1: return launch_drop<1>(P, grid, lds, s);
    case 3: return launch_drop<3>(P, grid, lds, s);
    case 4: return launch_drop<4>(P, grid, lds, s);
  }
  set_error("fused engine: no dropout kernel for K1=%d", K1);
  return PINN_ERR_UNSUPPORTED;
}

}  // namespace pinn
