// pinn_wide_w256.hip — instantiations of the wide (one launch per layer) MFMA engine, padded width 256
#include <type_traits>
#include "wide_kernel.h"

namespace pinn {

constexpr int NTW_ = 16;
constexpr size_t PADS_LDS = (size_t)(WIDE_WAVES * WIDE_MAX_PADS * TB_FLOATS + WIDE_WAVES * MAX_SUMS) * 4;

template <class K>
static int go(K kern, const FusedParams& P, const WideLayer& Lp, dim3 grid, size_t lds, hipStream_t s, const char* what) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return PINN_ERR_LAUNCH; }
  }
  hipLaunchKernelGGL(kern, grid, dim3(WIDE_THREADS), lds, s, P, Lp);
  return check_launch(what);
}

// 512-thread launch (two waves per 16-point tile: HV = 2 kernels)
template <class K>
static int go2(K kern, const FusedParams& P, const WideLayer& Lp, dim3 grid, size_t lds, hipStream_t s, const char* what) {
  hipLaunchKernelGGL(kern, grid, dim3(2 * WIDE_THREADS), lds, s, P, Lp);
  return check_launch(what);
}

template <int K1, bool BF>
// only the hidden (W x W) layers take bf16 operands; the thin first/last layers stay fp32
static int fwd_k(int which, bool grad, const FusedParams& P, const WideLayer& Lp, int grid, hipStream_t s) {
  constexpr int A = PINN_ACT_TANH;
  switch (which) {
    case 0:   // (bf16 mode: jets are stored as bf16 — wide_kernel.h FMT bits — while the thin layers' MFMAs stay fp32)
      if constexpr (BF) return go(k_wide_fwd<1, NTW_, K1, A, true, false, false, false, 1, FMT_OUT16>, P, Lp, dim3(grid), 0, s, "wide fwd first");
      else return go(k_wide_fwd<1, NTW_, K1, A, true, false, false, false>, P, Lp, dim3(grid), 0, s, "wide fwd first");
    case 1:   // two waves per tile pay off only once bf16 has shortened the MFMA block (measured: fp32 -8 %, bf16 +11 %)
      if constexpr (BF) return go2(k_wide_fwd<NTW_, NTW_, K1, A, false, false, false, true, 2, FMT_IN16 | FMT_OUT16>, P, Lp, dim3(grid), 0, s, "wide fwd hidden");
      else return go(k_wide_fwd<NTW_, NTW_, K1, A, false, false, false, false, 1>, P, Lp, dim3(grid), 0, s, "wide fwd hidden");
    default:
      if constexpr (BF)
        return grad ? go(k_wide_fwd<NTW_, 1, K1, A, false, true, true, false, 1, FMT_IN16>, P, Lp, dim3(grid), PADS_LDS, s, "wide fwd last")
                    : go(k_wide_fwd<NTW_, 1, K1, A, false, true, false, false, 1, FMT_IN16>, P, Lp, dim3(grid), PADS_LDS, s, "wide fwd last");
      else
        return grad ? go(k_wide_fwd<NTW_, 1, K1, A, false, true, true, false>, P, Lp, dim3(grid), PADS_LDS, s, "wide fwd last")
                    : go(k_wide_fwd<NTW_, 1, K1, A, false, true, false, false>, P, Lp, dim3(grid), PADS_LDS, s, "wide fwd last");
  }
}
template <>
int launch_wide_fwd<NTW_>(int which, int K1, int prec, bool grad, const FusedParams& P, const WideLayer& Lp, int grid, hipStream_t s) {
  const bool bf = prec == PINN_PREC_BF16;
  switch (K1) {
    case 1: return bf ? fwd_k<1, true>(which, grad, P, Lp, grid, s) : fwd_k<1, false>(which, grad, P, Lp, grid, s);
    case 3: return bf ? fwd_k<3, true>(which, grad, P, Lp, grid, s) : fwd_k<3, false>(which, grad, P, Lp, grid, s);
    case 4: return bf ? fwd_k<4, true>(which, grad, P, Lp, grid, s) : fwd_k<4, false>(which, grad, P, Lp, grid, s);
  }
  set_error("wide engine: no kernel for K1=%d", K1); return PINN_ERR_UNSUPPORTED;
}

template <int K1, bool BF>
static int bwd_k(int which, const FusedParams& P, const WideLayer& Lp, int grid, hipStream_t s) {
  constexpr int A = PINN_ACT_TANH;
  switch (which) {
    case 0:
      if constexpr (BF) return go(k_wide_bwd<NTW_, 1, K1, A, true, false, false, 1, FMT_IN16 | FMT_GIN16 | FMT_OUT16>, P, Lp, dim3(grid), 0, s, "wide bwd first");
      else return go(k_wide_bwd<NTW_, 1, K1, A, true, false, false>, P, Lp, dim3(grid), 0, s, "wide bwd first");
    case 1:
      if constexpr (BF) return go2(k_wide_bwd<NTW_, NTW_, K1, A, true, true, true, 2, FMT_IN16 | FMT_GIN16 | FMT_OUT16>, P, Lp, dim3(grid), 0, s, "wide bwd hidden");
      else return go(k_wide_bwd<NTW_, NTW_, K1, A, true, true, false, 1>, P, Lp, dim3(grid), 0, s, "wide bwd hidden");
    default:
      if constexpr (BF) return go(k_wide_bwd<1, NTW_, K1, A, false, true, false, 1, FMT_OUT16>, P, Lp, dim3(grid), 0, s, "wide bwd last");
      else return go(k_wide_bwd<1, NTW_, K1, A, false, true, false>, P, Lp, dim3(grid), 0, s, "wide bwd last");
  }
}
template <>
int launch_wide_bwd<NTW_>(int which, int K1, int prec, const FusedParams& P, const WideLayer& Lp, int grid, hipStream_t s) {
  const bool bf = prec == PINN_PREC_BF16;
  switch (K1) {
    case 1: return bf ? bwd_k<1, true>(which, P, Lp, grid, s) : bwd_k<1, false>(which, P, Lp, grid, s);
    case 3: return bf ? bwd_k<3, true>(which, P, Lp, grid, s) : bwd_k<3, false>(which, P, Lp, grid, s);
    case 4: return bf ? bwd_k<4, true>(which, P, Lp, grid, s) : bwd_k<4, false>(which, P, Lp, grid, s);
  }
  set_error("wide engine: no kernel for K1=%d", K1); return PINN_ERR_UNSUPPORTED;
}

template <int K1, bool BF>
static int wg_k(int which, const FusedParams& P, const WideLayer& Lp, int gx, hipStream_t s) {
  switch (which) {
    case 0: return go(k_wide_wgrad<4, NTW_, 1, K1, true, false, BF ? FMT_GIN16 : 0>, P, Lp, dim3(gx), PADS_LDS, s, "wide wgrad first");
    case 1: return go(k_wide_wgrad<4, NTW_, NTW_, K1, false, BF, BF ? (FMT_IN16 | FMT_GIN16) : 0>, P, Lp, dim3(gx), PADS_LDS, s, "wide wgrad hidden");
    default: return go(k_wide_wgrad<1, 1, NTW_, K1, false, false, BF ? FMT_IN16 : 0>, P, Lp, dim3(gx), PADS_LDS, s, "wide wgrad last");
  }
}
template <>
int launch_wide_wgrad<NTW_>(int which, int K1, int prec, const FusedParams& P, const WideLayer& Lp, int gx, hipStream_t s) {
  const bool bf = prec == PINN_PREC_BF16;
  switch (K1) {
    case 1: return bf ? wg_k<1, true>(which, P, Lp, gx, s) : wg_k<1, false>(which, P, Lp, gx, s);
    case 3: return bf ? wg_k<3, true>(which, P, Lp, gx, s) : wg_k<3, false>(which, P, Lp, gx, s);
    case 4: return bf ? wg_k<4, true>(which, P, Lp, gx, s) : wg_k<4, false>(which, P, Lp, gx, s);
  }
  set_error("wide engine: no kernel for K1=%d", K1); return PINN_ERR_UNSUPPORTED;
}

}  // namespace pinn
