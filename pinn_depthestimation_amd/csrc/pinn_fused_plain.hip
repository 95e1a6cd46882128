// pinn_fused_plain.hip — the PLAIN forward (pinn_forward: DNN.forward, dnn.py:54-55; test.py:76,96 evaluates grids of
// points with it), FOUR 16-point tiles per wave and pass (padded hidden width 16 / 32 / 64).
//
// k_fused<64, 1, false, ...> runs one 16-column GEMM per weight block: every 16-point tile re-streams each layer's 16 KB of
// weights (~20 TB/s of L2 reads chip-wide at 2^20 points per ms), and the weight loads' issue sits between 64-MFMA blocks.
// Here the four "quantity" slots of the jet kernel's GEMMs (gemm_chain / gemm_stream with K1 = 4) carry four different
// TILES instead of a value and three tangents: each weight block is fetched once per 64 points, the activation is a tanh
// on all four slots, and a wave's pass is 64 points.  Same arithmetic per point as the one-tile kernel (the fmaf chain of
// the fp32 MFMA over the same k order), so Y is bit-identical to it.
#include "fused_kernel.h"

namespace pinn {

constexpr int plain4_occ(int WP) { return WP == 64 ? 2 : 4; }     // workgroups per CU = waves per SIMD the instance is built for

template <int WP, int ACT>
__global__ __launch_bounds__(FUSED_THREADS, plain4_occ(WP)) void k_fused_plain4(const FusedParams P) {
  constexpr int NTH = WP / 16, C = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * FUSED_WAVES + wave, nw = (int64_t)gridDim.x * FUSED_WAVES;
  const float* __restrict__ Wp_ = P.Wp;
  const float* __restrict__ Bp_ = P.Bp;
  const int L = P.L;
  const int64_t n_groups = (P.n_tiles + C - 1) / C;
  for (int64_t g = gw; g < n_groups; g += nw) {
    int64_t pt[C];
    f4 b0[C][1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      pt[c] = (g * C + c) * 16 + p;
      const int64_t pc = pt[c] < P.N ? pt[c] : P.N - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * q + r;
        b0[c][0][r] = f < P.d_in ? P.X[pc * P.d_in + f] : 0.f;
      }
    }
    f4 a[C][NTH];
    f4 ws[NTH];   // the weight block the next GEMM starts with
    {
      f4 w0[NTH][1];
      load_w<1, NTH>(Wp_, w0, p, q);
      load_wblk<NTH>(Wp_ + w_off_p<WP>(L > 1 ? 1 : L), 0, ws, p, q);
      f4 bias[NTH];
      load_bias<NTH>(Bp_ + b_off_p<WP>(0), bias, q);
      f4 acc0[C][NTH];
      zero_tiles<NTH, C>(acc0);
      gemm_chain<1, NTH, C>(w0, b0, acc0);
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int MT = 0; MT < NTH; ++MT)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = acc0[c][MT][r] + bias[MT][r];
            a[c][MT][r] = ACT == PINN_ACT_TANH ? tanh_f32(z) : (z > 0.f ? z : 0.01f * z);
          }
    }
    for (int l = 1; l < L; ++l) {
      f4 bias[NTH];
      load_bias<NTH>(Bp_ + b_off_p<WP>(l), bias, q);
      f4 nx[C][NTH];
      zero_tiles<NTH, C>(nx);
      gemm_stream<NTH, NTH, C>(Wp_ + w_off_p<WP>(l), Wp_ + w_off_p<WP>(l + 1), ws, a, nx, p, q);
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int MT = 0; MT < NTH; ++MT)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = nx[c][MT][r] + bias[MT][r];
            a[c][MT][r] = ACT == PINN_ACT_TANH ? tanh_f32(z) : (z > 0.f ? z : 0.01f * z);
          }
    }
    f4 out[C][1];
    {
      f4 bias_o[1];
      load_bias<1>(Bp_ + b_off_p<WP>(L), bias_o, q);
      zero_tiles<1, C>(out);
      // (the block fetched behind the output GEMM is the next pass's first streamed one; it is re-requested at the top)
      gemm_stream<NTH, 1, C>(Wp_ + w_off_p<WP>(L), Wp_ + w_off_p<WP>(L > 1 ? 1 : L), ws, a, out, p, q);
#pragma unroll
      for (int c = 0; c < C; ++c) out[c][0] += bias_o[0];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (pt[c] < P.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 4 * q + r;
          if (o < P.d_out) P.Y[pt[c] * P.d_out + o] = out[c][0][r];
        }
      }
    }
  }
}

template <int WP>
static int launch_plain(const FusedParams& P, int64_t n_tiles, int cus, hipStream_t s) {
  const int64_t want = ((n_tiles + 3) / 4 + FUSED_WAVES - 1) / FUSED_WAVES, cap = (int64_t)plain4_occ(WP) * cus;
  const int grid = (int)(want < cap ? want : cap);
  if (P.act == PINN_ACT_TANH) hipLaunchKernelGGL((k_fused_plain4<WP, PINN_ACT_TANH>), dim3(grid), dim3(FUSED_THREADS), 0, s, P);
  else hipLaunchKernelGGL((k_fused_plain4<WP, PINN_ACT_LEAKY_RELU>), dim3(grid), dim3(FUSED_THREADS), 0, s, P);
  return check_launch("fused plain-forward kernel (four tiles per wave)");
}

// tiles below which the one-tile kernel spreads the points better: one pass of four tiles for every wave the chip holds
int64_t fused_plain_min_tiles(int WP, int cus) { return (int64_t)4 * FUSED_WAVES * plain4_occ(WP) * cus; }

int launch_fused_plain(int WP, const FusedParams& P, int cus, hipStream_t s) {
  return WP == 16 ? launch_plain<16>(P, P.n_tiles, cus, s) : WP == 32 ? launch_plain<32>(P, P.n_tiles, cus, s)
                                                                        : launch_plain<64>(P, P.n_tiles, cus, s);
}

}  // namespace pinn
