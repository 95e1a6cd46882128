// pinn_fused_coop.hip — instantiations of the cooperative (four waves per tile) fused kernel for
// small point sets (fused_coop_kernel.h); hidden width padded to 64
#include <type_traits>
#include "fused_coop_kernel.h"

namespace pinn {

template <int K1, bool GRAD, int ACT>
static int launch_coop_act(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  auto kern = k_fused_coop<K1, GRAD, ACT>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(COOP_THREADS), lds, s, P);
  return check_launch("fused cooperative kernel");
}

// residual-only gradient kernels with the epilogue specialised to one residual family (fused_kernel.h, EPI)
template <int K1, int EPI>
static int launch_coop_special(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  auto kern = k_fused_coop<K1, true, PINN_ACT_TANH, EPI>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(COOP_THREADS), lds, s, P);
  return check_launch("fused cooperative kernel (specialised epilogue)");
}

template <int K1, bool GRAD>
static int launch_coop(const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  if constexpr (GRAD && K1 >= 3) {
    if (P.act == PINN_ACT_TANH && P.loss_kind == 1 && P.Y == nullptr && P.n_split < 0) {
      if constexpr (K1 == 4) {
        if (P.residual_id == PINN_RES_NAVIER_STOKES) return launch_coop_special<4, EPI_NS>(P, grid, lds, s);
      }
      if constexpr (K1 == 3) {
        if (P.residual_id == PINN_RES_PHYSICS_EQUATION) return launch_coop_special<3, EPI_PE>(P, grid, lds, s);
        if (P.residual_id == PINN_RES_CONTINUITY_ONLY || P.residual_id == PINN_RES_CONTINUITY_FTEMP)
          return launch_coop_special<3, EPI_CONT>(P, grid, lds, s);
      }
    }
  }
  return P.act == PINN_ACT_TANH ? launch_coop_act<K1, GRAD, PINN_ACT_TANH>(P, grid, lds, s)
                                : launch_coop_act<K1, GRAD, PINN_ACT_LEAKY_RELU>(P, grid, lds, s);
}

int launch_fused_coop(int K1, bool grad, const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  if (!grad) {
    switch (K1) {
      case 1: return launch_coop<1, false>(P, grid, lds, s);
      case 2: return launch_coop<2, false>(P, grid, lds, s);
      case 3: return launch_coop<3, false>(P, grid, lds, s);
      case 4: return launch_coop<4, false>(P, grid, lds, s);
    }
  } else {
    switch (K1) {
      case 1: return launch_coop<1, true>(P, grid, lds, s);
      case 3: return launch_coop<3, true>(P, grid, lds, s);
      case 4: return launch_coop<4, true>(P, grid, lds, s);
    }
  }
  set_error("fused cooperative engine: no kernel for K1=%d grad=%d", K1, (int)grad);
  return PINN_ERR_UNSUPPORTED;
}

}  // namespace pinn
