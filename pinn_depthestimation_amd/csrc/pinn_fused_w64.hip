// pinn_fused_w64.hip — instantiations of the fused MFMA chain kernel for padded hidden width 64
#include <type_traits>
#include "fused_kernel.h"

namespace pinn {

template <int K1, bool GRAD, bool LDSACC, int ACT>
static int launch_one_act(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  auto kern = k_fused<64, K1, GRAD, LDSACC, ACT>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), lds, s, P);
  return check_launch("fused kernel (WP=64)");
}

// residual-only gradient kernels with the epilogue specialised to one residual family (fused_kernel.h, EPI)
template <int K1, int EPI>
static int launch_special(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  // k-step-major inputs / outputs (fused_kernel.h, KRO): ceil(d_out / 4) k-steps in the output layer's reverse GEMM
  constexpr int KRO = EPI == EPI_PE ? 2 : 1;
  auto kern = k_fused<64, K1, true, true, PINN_ACT_TANH, EPI, KRO>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), lds, s, P);
  return check_launch("fused kernel (WP=64, specialised epilogue)");
}

template <int K1, bool GRAD, bool LDSACC>
static int launch_one(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  if constexpr (GRAD && LDSACC && K1 >= 3) {
    if (P.io1) {   // (pinn_fused.hip decides: tanh, residual only, no outputs wanted, d_in <= 4, d_out within the epilogue's k-steps)
      if constexpr (K1 == 4) {
        if (P.residual_id == PINN_RES_NAVIER_STOKES) return launch_special<4, EPI_NS>(P, grid, lds, s);
      }
      if constexpr (K1 == 3) {
        if (P.residual_id == PINN_RES_PHYSICS_EQUATION) return launch_special<3, EPI_PE>(P, grid, lds, s);
        if (P.residual_id == PINN_RES_CONTINUITY_ONLY || P.residual_id == PINN_RES_CONTINUITY_FTEMP)
          return launch_special<3, EPI_CONT>(P, grid, lds, s);
      }
    }
  }
  return P.act == PINN_ACT_TANH ? launch_one_act<K1, GRAD, LDSACC, PINN_ACT_TANH>(P, grid, lds, s)
                                : launch_one_act<K1, GRAD, LDSACC, PINN_ACT_LEAKY_RELU>(P, grid, lds, s);
}

template <>
int launch_fused<64>(int K1, bool grad, const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  if (!grad) {
    switch (K1) {
      case 1: return launch_one<1, false, false>(P, grid, lds, s);
      case 2: return launch_one<2, false, false>(P, grid, lds, s);
      case 3: return launch_one<3, false, false>(P, grid, lds, s);
      case 4: return launch_one<4, false, false>(P, grid, lds, s);
    }
  } else {
    switch (K1) {
      case 1: return P.acc_lds ? launch_one<1, true, true>(P, grid, lds, s) : launch_one<1, true, false>(P, grid, lds, s);
      case 3: return P.acc_lds ? launch_one<3, true, true>(P, grid, lds, s) : launch_one<3, true, false>(P, grid, lds, s);
      case 4: return P.acc_lds ? launch_one<4, true, true>(P, grid, lds, s) : launch_one<4, true, false>(P, grid, lds, s);
    }
  }
  set_error("fused engine: no kernel for K1=%d grad=%d", K1, (int)grad);
  return PINN_ERR_UNSUPPORTED;
}

}  // namespace pinn
