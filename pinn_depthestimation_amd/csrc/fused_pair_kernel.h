// fused_pair_kernel.h — the fused chain kernel (fused_kernel.h) re-tiled for TWO waves per SIMD.
//
// k_fused keeps a 16-point x (1+k)-quantity tile per wave, which needs the whole 512-register
// budget: one wave per SIMD, so every non-MFMA instruction (tanh, adjoints, transposes, spills,
// the LDS gradient flush) leaves the matrix pipe idle (68 % MFMA-busy measured).  Here a wave
// owns 8 points and the 16 MFMA columns carry 8 points x 2 quantities:
//     column n = lane&15:  point n&7,  quantity 2*set + (n>>3)      (set = register set)
// so an even jet (1+k in {2,4}) needs (1+k)/2 register sets, every per-wave array halves, a wave
// fits 256 registers, and a second wave on the same SIMD fills the matrix pipe while this one
// does its vector work.  The chain / K-permutation / weight-gradient tricks are unchanged (they
// are column-agnostic); what differs is everything that couples quantities of one point:
//   - activation: the tangent columns need s = 1 - tanh(z)^2 of the primal column 8 lanes away
//     (one DPP row_ror:8, folded into the multiply);
//   - activation adjoint: the primal column needs the tangent columns' products (same DPP);
//   - residual gather / adjoint scatter and the input jet index points by lane&7.
// Same arithmetic as k_fused up to the summation order of the tanh'' cross term.
#pragma once
#include "fused_kernel.h"

namespace pinn {

#ifndef PINN_PR_WAVES
#define PINN_PR_WAVES 8
#endif
constexpr int PR_WAVES = PINN_PR_WAVES;
constexpr int PR_THREADS = PR_WAVES * 64;
constexpr int PR_TB_PER_WAVE = 4;
#ifndef PR_OPT_PIPE
#define PR_OPT_PIPE 1
#endif
#ifndef PR_OPT_BATCH
#define PR_OPT_BATCH 1
#endif   // 4 KB of pads per wave: 125 KB of LDS gradient + 8 waves must fit 160 KB

__device__ __forceinline__ float dpp_ror8(float v) {   // value of the lane 8 columns away in this row of 16
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}


template <int NT, int NS>
__device__ __forceinline__ void init_bias_pr(const float* __restrict__ b, f4 (&acc)[NS][NT], int q, int j) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT) {
    const f4 bv = *reinterpret_cast<const f4*>(b + 16 * MT + 4 * q);
    acc[0][MT] = j ? f4{0.f, 0.f, 0.f, 0.f} : bv;
#pragma unroll
    for (int i = 1; i < NS; ++i) acc[i][MT] = f4{0.f, 0.f, 0.f, 0.f};
  }
}

template <int ACT, int NT, int NS>
__device__ __forceinline__ void activate_pr(f4 (&acc)[NS][NT], int j) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = acc[0][MT][r];
      float a, s_own;
      if constexpr (ACT == PINN_ACT_TANH) { a = tanh_f32(z); s_own = fmaf(-a, a, 1.f); }
      else { a = z > 0.f ? z : 0.01f * z; s_own = z > 0.f ? 1.f : 0.01f; }
      const float s_oth = dpp_ror8(s_own);
      const float s = j ? s_oth : s_own;
      acc[0][MT][r] = j ? z * s : a;
#pragma unroll
      for (int i = 1; i < NS; ++i) acc[i][MT][r] *= s;
    }
}

template <int ACT, int NT, int NS>
__device__ __forceinline__ void activate_adjoint_pr(f4 (&G)[NS][NT], const f4 (&A)[NS][NT], int j) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a0 = A[0][MT][r], g0 = G[0][MT][r];
      const float a_oth = dpp_ror8(a0);
      const float a = j ? a_oth : a0;   // the point's primal activation
      if constexpr (ACT == PINN_ACT_TANH) {
        const float s = fmaf(-a, a, 1.f);
        float part = j ? g0 * a0 : 0.f;
#pragma unroll
        for (int i = 1; i < NS; ++i) {
          part = fmaf(G[i][MT][r], A[i][MT][r], part);
          G[i][MT][r] *= s;
        }
        const float cross = part + dpp_ror8(part);
        G[0][MT][r] = j ? g0 * s : fmaf(-2.f * a, cross, s * g0);   // tanh'' = -2 a (1 - a^2)
      } else {
        const float s = a > 0.f ? 1.f : 0.01f;
#pragma unroll
        for (int i = 0; i < NS; ++i) G[i][MT][r] *= s;
      }
    }
}

// dW[16MT + 4q + r][16NT + n] += sum_set sum_columns Z[set][MT] * A[set][NT];  db from the primal columns
template <int MT_N, int NT_N, int NS, class Sink>
__device__ __forceinline__ void weight_grad_pr(const Sink& sink, int layer, int woff, int boff, const f4 (&Z)[NS][MT_N],
                                               const f4 (&A)[NS][NT_N], float* __restrict__ tb, int lane) {
  const int p = lane & 15, q = lane >> 4;
  f4 dw[MT_N][NT_N];
  float bs[MT_N];
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) dw[MT][NT] = f4{0.f, 0.f, 0.f, 0.f};
  // The four pads are used twice per set (Z blocks, then A blocks); one wave's LDS operations
  // execute in order, so no barrier is needed.  Set i+1's round trips are issued before set i's
  // MFMAs and land while they run.
  f4 zt[2][MT_N], at[2][NT_N];
  auto stage = [&](int i, f4 (&z)[MT_N], f4 (&a)[NT_N]) {
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) transpose_write(tb + MT * TB_FLOATS, Z[i][MT], p, q);
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) z[MT] = transpose_read(tb + MT * TB_FLOATS, p, q);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) transpose_write(tb + NT * TB_FLOATS, A[i][NT], p, q);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) a[NT] = transpose_read(tb + NT * TB_FLOATS, p, q);
  };
  stage(0, zt[0], at[0]);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    if (PR_OPT_PIPE && i + 1 < NS) stage(i + 1, zt[(i + 1) & 1], at[(i + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    if (i == 0) {   // zt[MT][s] = zbar(feature p, column 4s + q); columns 0..7 (s < 2) are the primal
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT) {
        float t = zt[0][MT][0] + zt[0][MT][1];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        bs[MT] = t;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
        for (int NT = 0; NT < NT_N; ++NT) dw[MT][NT] = mfma4(zt[i & 1][MT][s], at[i & 1][NT][s], dw[MT][NT]);
    if (!PR_OPT_PIPE && i + 1 < NS) stage(i + 1, zt[(i + 1) & 1], at[(i + 1) & 1]);
  }
  // flush: one LDS round trip per row block (reads batched, then adds + writes), not one per tile
  sink.lock(layer, lane);
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT) {
    f4 cur[NT_N];
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) cur[NT] = *reinterpret_cast<const f4*>(sink.acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4);
    if (PR_OPT_BATCH) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) *reinterpret_cast<f4*>(sink.acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4) = cur[NT] + dw[MT][NT];
  }
  if (q == 0) {
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) sink.add1(boff + 16 * MT + p, bs[MT]);
  }
  sink.unlock(layer, lane);
}

template <int NS>
struct ScatterMapPr {
  int role_of[4];
  int cinv[NS];
};

template <int NS>
__device__ __forceinline__ void build_scatter_maps_pr(const FusedParams& P, int q, int j, ScatterMapPr<NS>& sm,
                                                      ScatterMapPr<NS>& sm_mse) {
#pragma unroll
  for (int r2 = 0; r2 < 4; ++r2) {
    int ro = -1, rm = -1;
    for (int r = PINN_MAX_ROLES - 1; r >= 0; --r) {
      ro = (P.out_col[r] == 4 * q + r2) ? r : ro;
      rm = (r < P.n_cols && P.mse_col[r] == 4 * q + r2) ? r : rm;
    }
    sm.role_of[r2] = ro;
    sm_mse.role_of[r2] = rm;
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int ce = 2 * i + j;   // this lane's quantity in register set i
    int ci = (ce == 0) ? 0 : -1;
    for (int d = PINN_MAX_DIRS - 1; d >= 0; --d) ci = (ce > 0 && P.q_of[d] == ce) ? 1 + d : ci;
    sm.cinv[i] = ci;
    sm_mse.cinv[i] = (ce == 0) ? 0 : -1;
  }
}

// output column o of quantity c for this lane's point p8
template <int NS>
__device__ __forceinline__ float gather_out_pr(const f4 (&out)[NS][1], int c, int o, int p8) {
  f4 tile = out[0][0];
#pragma unroll
  for (int i = 1; i < NS; ++i) tile = ((c >> 1) == i) ? out[i][0] : tile;
  return __shfl(pick4(tile, o & 3), p8 + 8 * (c & 1) + 16 * (o >> 2), 64);
}

template <int NS, int NC, int NR, bool ACCUM = false>
__device__ __forceinline__ void scatter_adjoint_pr(float* __restrict__ tb, const float (&g)[NC][NR],
                                                   const ScatterMapPr<NS>& sm, f4 (&G)[NS][1], bool valid, int lane) {
  const int p8 = lane & 7;
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < NR; ++r) tb[(c * NR + r) * 8 + p8] = g[c][r];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
      const int ci = sm.cinv[i], ro = sm.role_of[r2];
      const bool ok = valid && ci >= 0 && ci < NC && ro >= 0;
      const float val = tb[ok ? (ci * NR + ro) * 8 + p8 : p8];
      G[i][0][r2] = (ACCUM ? G[i][0][r2] : 0.f) + (ok ? val : 0.f);
    }
  __builtin_amdgcn_wave_barrier();
}

template <class RES, int NS, bool GRAD>
__device__ __forceinline__ void residual_tile_pr(const FusedParams& P, const f4 (&out)[NS][1], f4 (&G)[NS][1],
                                                 float (&sums)[MAX_SUMS], const ScatterMapPr<NS>& sm,
                                                 float* __restrict__ tb, bool valid, bool masked, int lane) {
  constexpr int NR = RES::NR, ND = RES::ND, NT = RES::NT;
  const int p8 = lane & 7;
  float v[1 + ND][NR], g[1 + ND][NR], sq[NT], sc[NT];
#pragma unroll
  for (int c = 0; c <= ND; ++c) {
    const int ce = (c == 0) ? 0 : P.q_of[c - 1];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[c][r] = gather_out_pr<NS>(out, ce, P.out_col[r], p8);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) sc[t] = GRAD ? P.scale[t] : 0.f;
  if constexpr (std::is_same<RES, ResContinuity>::value)
    RES::template eval<GRAD>(v, sc, g, sq, P.residual_id == PINN_RES_CONTINUITY_ONLY, masked, P.anchor);
  else
    RES::template eval<GRAD>(v, sc, g, sq);
  if (valid && lane < 8) {
#pragma unroll
    for (int t = 0; t < NT; ++t) sums[t] += sq[t];
  }
  if constexpr (GRAD) scatter_adjoint_pr<NS, 1 + ND, NR>(tb, g, sm, G, valid, lane);
}

template <int K1, bool GRAD>
__device__ __forceinline__ void loss_epilogue_pr(const FusedParams& P, const f4 (&out)[K1 / 2][1], f4 (&G)[K1 / 2][1],
                                                 float (&sums)[MAX_SUMS], const ScatterMapPr<K1 / 2>& sm,
                                                 const ScatterMapPr<K1 / 2>& sm_mse, float* __restrict__ tb, int64_t pt,
                                                 int64_t ptc, bool valid, int lane) {
  constexpr int NS = K1 / 2;
  const int q = lane >> 4, j = (lane >> 3) & 1, p8 = lane & 7;
  if (P.Y != nullptr && valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      if (o < P.d_out) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int c = 2 * i + j;
          if (c == 0) P.Y[pt * P.d_out + o] = out[i][0][r];
          else if (P.dY != nullptr) P.dY[((int64_t)(c - 1) * P.N + pt) * P.d_out + o] = out[i][0][r];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) G[i][0] = f4{0.f, 0.f, 0.f, 0.f};
  if (P.loss_kind & 1) {
    if (P.residual_id == PINN_RES_NAVIER_STOKES) {
      if constexpr (K1 >= 4) residual_tile_pr<ResNavierStokes, NS, GRAD>(P, out, G, sums, sm, tb, valid, false, lane);
    } else if (P.residual_id == PINN_RES_PHYSICS_EQUATION) {
      if constexpr (K1 >= 3) residual_tile_pr<ResPhysicsEquation, NS, GRAD>(P, out, G, sums, sm, tb, valid, false, lane);
    } else {
      if constexpr (K1 >= 3) {
        const bool masked = P.residual_id == PINN_RES_CONTINUITY_ONLY && P.X[ptc * P.d_in + P.xcol] < P.thr;
        residual_tile_pr<ResContinuity, NS, GRAD>(P, out, G, sums, sm, tb, valid, masked, lane);
      }
    }
  }
  if (P.loss_kind & 2) {
    float gm[1][PINN_MAX_ROLES];
#pragma unroll
    for (int jc = 0; jc < PINN_MAX_ROLES; ++jc) {
      gm[0][jc] = 0.f;
      if (jc < P.n_cols) {
        const float y = gather_out_pr<NS>(out, 0, P.mse_col[jc], p8);
        const float d = P.T[ptc * P.n_cols + jc] - y;                 // train.py:141 (true - pred)
        if (valid && lane < 8) sums[MSE_SUM0 + jc] += d * d;
        if (GRAD) gm[0][jc] = -2.f * P.mse_scale[jc] * d;
      }
    }
    if constexpr (GRAD) scatter_adjoint_pr<NS, 1, PINN_MAX_ROLES, true>(tb, gm, sm_mse, G, valid, lane);
  }
}

// One workgroup per CU, 8 waves = 2 per SIMD; the LDS gradient copy is shared by all 8.
template <int WP, int K1, bool GRAD, int ACT>
__global__ __launch_bounds__(PR_THREADS, PR_WAVES / 4) void k_fused_pair(const FusedParams P) {
  static_assert(K1 == 2 || K1 == 4, "paired layout needs an even jet");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTH = WP / 16, NS = K1 / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4, j = (lane >> 3) & 1, p8 = lane & 7;
  float* lacc = smem;
  int* locks = reinterpret_cast<int*>(smem + P.lds_acc_floats);
  float* tb = smem + P.lds_acc_floats + MAX_LOCKS + wave * (PR_TB_PER_WAVE * TB_FLOATS);
  float* lsum = smem + P.lds_acc_floats + MAX_LOCKS + PR_WAVES * PR_TB_PER_WAVE * TB_FLOATS;
  const int PP = P.PW + P.PB;
  GradSink<true> sink;
  sink.acc = lacc;
  sink.locks = locks;
  if (GRAD) {
    for (int i = threadIdx.x; i < PP; i += PR_THREADS) lacc[i] = 0.f;
    if (threadIdx.x < MAX_LOCKS) locks[threadIdx.x] = 0;
    __syncthreads();
  }
  float sums[MAX_SUMS];
#pragma unroll
  for (int t = 0; t < MAX_SUMS; ++t) sums[t] = 0.f;

  ScatterMapPr<NS> sm, sm_mse;
  build_scatter_maps_pr<NS>(P, q, j, sm, sm_mse);
  const int gw = blockIdx.x * PR_WAVES + wave, nw = gridDim.x * PR_WAVES;
  float* __restrict__ scr = P.scratch + (int64_t)gw * P.scratch_per_wave;
  const float* __restrict__ Wp_ = P.Wp;
  const float* __restrict__ WTp_ = P.WTp;
  const float* __restrict__ Bp_ = P.Bp;
  constexpr int SLOT = NS * NTH * 256;  // floats per spilled layer
  const int L = P.L;

  for (int64_t tile = gw; tile < P.n_tiles; tile += nw) {
    const int64_t pt = tile * 8 + p8;
    const bool valid = pt < P.N;
    const int64_t ptc = valid ? pt : P.N - 1;
    // ---- layer-0 input jet: features 4q + r of (x | unit tangent) for this lane's (point, quantity) ----
    auto input_jet = [&](f4 (&b)[NS][1]) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int dc = (i == 0) ? (j ? P.dir_col[0] : -1) : (j ? P.dir_col[2 * i] : P.dir_col[2 * i - 1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 4 * q + r;
          const float x = (f < P.d_in) ? P.X[ptc * P.d_in + f] : 0.f;
          b[i][0][r] = (i == 0 && j == 0) ? x : ((f == dc) ? 1.f : 0.f);
        }
      }
    };
    f4 b0[NS][1];
    input_jet(b0);
    // ---- forward chain -------------------------------------------------------------------------
    f4 a[NS][NTH];
    f4 ws[NTH];   // the streamed weight block that the next GEMM starts with
    {
      f4 w0[NTH][1];
      load_w<1, NTH>(Wp_, w0, p, q);
      load_wblk<NTH>(Wp_ + w_off_p<WP>(L > 1 ? 1 : L), 0, ws, p, q);
      init_bias_pr<NTH, NS>(Bp_ + b_off_p<WP>(0), a, q, j);
      gemm_chain<1, NTH, NS>(w0, b0, a);
    }
    activate_pr<ACT, NTH, NS>(a, j);
    if (GRAD && L > 1) spill<NTH, NS>(scr, a, lane);
    for (int l = 1; l < L; ++l) {
      f4 nx[NS][NTH];
      init_bias_pr<NTH, NS>(Bp_ + b_off_p<WP>(l), nx, q, j);
      gemm_stream<NTH, NTH, NS>(Wp_ + w_off_p<WP>(l), Wp_ + w_off_p<WP>(l + 1), ws, a, nx, p, q);
      activate_pr<ACT, NTH, NS>(nx, j);
      if (GRAD && l < L - 1) spill<NTH, NS>(scr + l * SLOT, nx, lane);
      copy_tiles<NTH, NS>(a, nx);
    }
    f4 out[NS][1];
    init_bias_pr<1, NS>(Bp_ + b_off_p<WP>(L), out, q, j);
    // (the block fetched behind the output GEMM is W_{L-1}^T's first block: the reverse sweep starts there)
    gemm_stream<NTH, 1, NS>(Wp_ + w_off_p<WP>(L), WTp_ + w_off_p<WP>(L > 1 ? L - 1 : 0), ws, a, out, p, q);
    f4 ai[NS][NTH];
    f4 wtl[NTH][1];
    if constexpr (GRAD) {
      unspill<NTH, NS>(scr + (L > 1 ? L - 2 : 0) * SLOT, ai, lane);   // a_{L-1}
      load_w<1, NTH>(WTp_ + w_off_p<WP>(L), wtl, p, q);
    }

    f4 G[NS][1];
    loss_epilogue_pr<K1, GRAD>(P, out, G, sums, sm, sm_mse, tb, pt, ptc, valid, lane);

    // ---- reverse sweep (structure of k_fused; the partner wave on this SIMD hides the latencies) ----
    if constexpr (GRAD) {
      weight_grad_pr<1, NTH, NS>(sink, L, w_off_p<WP>(L), P.PW + b_off_p<WP>(L), G, a, tb, lane);
      f4 g[NS][NTH];
      zero_tiles<NTH, NS>(g);
      gemm_chain<1, NTH, NS>(wtl, G, g);
      f4 ao[NS][NTH];
      copy_tiles<NTH, NS>(ao, a);
      for (int l = L - 1; l >= 1; --l) {
        // hidden layer l: output a_{l+1} (= ao), input a_l (= ai); ws holds W_l^T's first block
        activate_adjoint_pr<ACT, NTH, NS>(g, ao, j);
        weight_grad_pr<NTH, NTH, NS>(sink, l, w_off_p<WP>(l), P.PW + b_off_p<WP>(l), g, ai, tb, lane);
        f4 an[NS][NTH];
        unspill<NTH, NS>(scr + (l >= 2 ? l - 2 : 0) * SLOT, an, lane);            // a_{l-1} for the next iteration
        f4 g2[NS][NTH];
        zero_tiles<NTH, NS>(g2);
        gemm_stream<NTH, NTH, NS>(WTp_ + w_off_p<WP>(l), WTp_ + w_off_p<WP>(l >= 2 ? l - 1 : 1), ws, g, g2, p, q);
        copy_tiles<NTH, NS>(g, g2);
        copy_tiles<NTH, NS>(ao, ai);
        copy_tiles<NTH, NS>(ai, an);
      }
      {  // layer 0: output a_1 (= ao), input = (x, unit tangents)
        activate_adjoint_pr<ACT, NTH, NS>(g, ao, j);
        f4 b1[NS][1];
        input_jet(b1);
        weight_grad_pr<NTH, 1, NS>(sink, 0, 0, P.PW + b_off_p<WP>(0), g, b1, tb, lane);
      }
    }
  }

  // ---- per-workgroup reductions ------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < MAX_SUMS; ++t) {
    float v = sums[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) lsum[wave * MAX_SUMS + t] = v;
  }
  __syncthreads();
  if (threadIdx.x < MAX_SUMS) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < PR_WAVES; ++w) v += lsum[w * MAX_SUMS + threadIdx.x];
    P.wg_sums[(int64_t)blockIdx.x * MAX_SUMS + threadIdx.x] = v;
  }
  if (GRAD) {
    float* dst = P.wg_grads + (int64_t)blockIdx.x * PP;
    for (int i = threadIdx.x; i < PP; i += PR_THREADS) dst[i] = lacc[i];
  }
}

int launch_fused_pair(int K1, bool grad, const FusedParams& P, int grid, size_t lds_bytes, hipStream_t s);

}  // namespace pinn
