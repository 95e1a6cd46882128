// fused_coop_kernel.h — latency form of the fused chain kernel for SMALL point sets.
//
// k_fused gives a whole 16-point tile to one wave: the tile's forward + reverse chain is ~5700
// dependent MFMAs, ~110 us, however few points there are — and the reference's own problem sizes
// (N_res = 243, N_fid = 12, config_CMB.json:43; 9600 in config.json:35) are a handful of tiles.
// Here the FOUR waves of a workgroup share one tile: wave w owns feature block w (rows
// 16w..16w+15) of every hidden layer, so each GEMM is split four ways and a tile takes about a
// third of the time.  What it costs: the B operand of every GEMM is the FULL activation (or
// adjoint) jet, so each layer exchanges the four blocks through LDS (one barrier per layer
// forward, two backward).
//
// Layouts are those of fused_kernel.h.  Exchange buffers hold one 16x16 block per (quantity,
// feature block) in the XOR-swizzled pad layout of transpose_write(): the same bytes are read
// back as an accumulator-layout f4 (B operand) or, transposed, as the weight-gradient operands.
// Each wave accumulates ITS rows of dW/db in the workgroup's LDS gradient copy: rows are owned
// exclusively, so no lock.  Spilled activations: each wave spills and re-reads only its own block.
#pragma once
#include "fused_kernel.h"

namespace pinn {

constexpr int COOP_WAVES = 4;
constexpr int COOP_THREADS = 256;

__device__ __forceinline__ f4 pad_read_acc(const float* __restrict__ tb, int p, int q) {   // inverse of transpose_write
  return *reinterpret_cast<const f4*>(tb + p * 16 + 4 * (q ^ (p & 3)));
}

template <int K1>
__device__ __forceinline__ void coop_write_blocks(float* __restrict__ X, int w, const f4 (&v)[K1], int p, int q) {
#pragma unroll
  for (int c = 0; c < K1; ++c) transpose_write(X + (c * 4 + w) * TB_FLOATS, v[c], p, q);
}
template <int K1>
__device__ __forceinline__ void coop_read_full(const float* __restrict__ X, f4 (&a)[K1][4], int p, int q) {
#pragma unroll
  for (int c = 0; c < K1; ++c)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) a[c][kt] = pad_read_acc(X + (c * 4 + kt) * TB_FLOATS, p, q);
}

template <int ACT, int K1>
__device__ __forceinline__ void coop_activate(const f4 (&acc)[K1], f4 bias, f4 (&a)[K1]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float z = acc[0][r] + bias[r];
    float av, s;
    if constexpr (ACT == PINN_ACT_TANH) { av = tanh_f32(z); s = fmaf(-av, av, 1.f); }
    else { av = z > 0.f ? z : 0.01f * z; s = z > 0.f ? 1.f : 0.01f; }
    a[0][r] = av;
#pragma unroll
    for (int c = 1; c < K1; ++c) a[c][r] = acc[c][r] * s;
  }
}
template <int ACT, int K1>
__device__ __forceinline__ void coop_adjoint(const f4 (&G)[K1], const f4 (&A)[K1], f4 (&Z)[K1]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = A[0][r];
    if constexpr (ACT == PINN_ACT_TANH) {
      const float s = fmaf(-a, a, 1.f);
      float cross = 0.f;
#pragma unroll
      for (int c = 1; c < K1; ++c) {
        cross = fmaf(G[c][r], A[c][r], cross);
        Z[c][r] = G[c][r] * s;
      }
      Z[0][r] = fmaf(-2.f * a, cross, s * G[0][r]);
    } else {
      const float s = a > 0.f ? 1.f : 0.01f;
#pragma unroll
      for (int c = 0; c < K1; ++c) Z[c][r] = G[c][r] * s;
    }
  }
}

// One workgroup (4 waves) per 16-point tile; hidden width padded to 64.
template <int K1, bool GRAD, int ACT, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(COOP_THREADS, 1) void k_fused_coop(const FusedParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WP = 64, NTH = 4;
  constexpr int XF = K1 * 4 * TB_FLOATS;          // floats of one exchange buffer
  constexpr int SLOTB = K1 * 256;                 // floats one wave spills per layer (its own block)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  float* lacc = smem;
  float* XA = smem + P.lds_acc_floats;
  float* XZ = XA + XF;
  float* lsum = XZ + XF;
  const int PP = P.PW + P.PB;
  if (GRAD) {
    for (int i = threadIdx.x; i < PP; i += COOP_THREADS) lacc[i] = 0.f;
    __syncthreads();
  }
  float sums[MAX_SUMS];
#pragma unroll
  for (int j = 0; j < MAX_SUMS; ++j) sums[j] = 0.f;
  ScatterMap<K1> sm, sm_mse;
  build_scatter_maps<K1>(P, q, sm, sm_mse);
  const bool primary = (w == 0);
  float* __restrict__ scr = P.scratch + (int64_t)blockIdx.x * P.scratch_per_wave + w * SLOTB;   // + l * 4 * SLOTB
  const float* __restrict__ Wp_ = P.Wp;
  const float* __restrict__ WTp_ = P.WTp;
  const float* __restrict__ Bp_ = P.Bp;
  const int L = P.L;
  float* priv = XA + w * (XF / 2);                // wave-private pads (2*K1 of them) inside XA|XZ while both are idle

  for (int64_t tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    const int64_t pt = tile * 16 + p;
    const bool valid = pt < P.N;
    const int64_t ptc = valid ? pt : P.N - 1;
    auto input_jet = [&](f4 (&b)[K1][1]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * q + r;
        b[0][0][r] = (f < P.d_in) ? P.X[ptc * P.d_in + f] : 0.f;
#pragma unroll
        for (int c = 1; c < K1; ++c) b[c][0][r] = (f == P.dir_col[c - 1]) ? 1.f : 0.f;
      }
    };
    // ---- forward -------------------------------------------------------------------------------------
    f4 ablk[K1];          // this wave's block of the current activation jet
    f4 a[K1][NTH];        // the full jet (B operand)
    {
      f4 b0[K1][1];
      input_jet(b0);
      const f4 w0 = *reinterpret_cast<const f4*>(Wp_ + (16 * w + p) * 16 + 4 * q);
      const f4 bias = *reinterpret_cast<const f4*>(Bp_ + b_off_p<WP>(0) + 16 * w + 4 * q);
      f4 acc[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < K1; ++c) acc[c] = mfma4(w0[r], b0[c][0][r], acc[c]);
      coop_activate<ACT, K1>(acc, bias, ablk);
    }
    int cur = 0;   // exchange buffer the next write goes to (ping-pong forward: one barrier per layer)
    for (int l = 1; l <= L; ++l) {
      // publish a_l's blocks, fetch layer l's weights meanwhile (hidden: 16 rows x 64; output: split-K slice)
      float* X = cur ? XZ : XA;
      if (GRAD && l < L) {
#pragma unroll
        for (int c = 0; c < K1; ++c) *reinterpret_cast<f4*>(scr + (l - 1) * 4 * SLOTB + c * 256 + lane * 4) = ablk[c];
      }
      coop_write_blocks<K1>(X, w, ablk, p, q);
      f4 wb[NTH];
      f4 bias;
      if (l < L) {
        load_wblk<NTH>(Wp_ + w_off_p<WP>(l), w, wb, p, q);
        bias = *reinterpret_cast<const f4*>(Bp_ + b_off_p<WP>(l) + 16 * w + 4 * q);
      }
      __syncthreads();
      coop_read_full<K1>(X, a, p, q);
      cur ^= 1;
      if (l == L) break;
      f4 acc[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NTH; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < K1; ++c) acc[c] = mfma4(wb[kt][r], a[c][kt][r], acc[c]);
      coop_activate<ACT, K1>(acc, bias, ablk);
    }
    // ablk = own block of a_L, a = full a_L.  Output layer, split-K: wave w contracts features 16w..16w+15
    f4 out[K1][1];
    {
      float* X = cur ? XZ : XA;      // (the buffer a_L was NOT published in)
      const f4 wl = *reinterpret_cast<const f4*>(Wp_ + w_off_p<WP>(L) + p * WP + 16 * w + 4 * q);
      f4 part[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) part[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < K1; ++c) part[c] = mfma4(wl[r], a[c][w][r], part[c]);
#pragma unroll
      for (int c = 0; c < K1; ++c) *reinterpret_cast<f4*>(X + (c * 4 + w) * TB_FLOATS + lane * 4) = part[c];
      const f4 bias_o = *reinterpret_cast<const f4*>(Bp_ + b_off_p<WP>(L) + 4 * q);
      __syncthreads();
#pragma unroll
      for (int c = 0; c < K1; ++c) {
        f4 s = *reinterpret_cast<const f4*>(X + (c * 4 + 0) * TB_FLOATS + lane * 4);
#pragma unroll
        for (int k = 1; k < 4; ++k) s += *reinterpret_cast<const f4*>(X + (c * 4 + k) * TB_FLOATS + lane * 4);
        out[c][0] = s;
      }
      out[0][0] += bias_o;
      __syncthreads();               // partials consumed: both exchange buffers are free again
    }
    // reverse-sweep operands whose latency the loss evaluation hides
    f4 ai[K1];
    f4 wt[NTH];
    if constexpr (GRAD) {
      if (L > 1) {
#pragma unroll
        for (int c = 0; c < K1; ++c) ai[c] = *reinterpret_cast<const f4*>(scr + (L - 2) * 4 * SLOTB + c * 256 + lane * 4);   // a_{L-1}
        load_wblk<NTH>(WTp_ + w_off_p<WP>(L - 1), w, wt, p, q);
      }
    }

    // ---- outputs / loss: every wave evaluates it (each needs the output adjoint); wave 0 stores and sums ----
    f4 G[K1][1];
    loss_epilogue<K1, GRAD, true, EPI>(P, out, G, sums, sm, sm_mse, priv, pt, ptc, valid, p, q, primary);

    if constexpr (GRAD) {
      // ---- output layer L: dW_L columns 16w.., abar_L block w, zbar_{L-1} block w ------------------------
      {
        f4 dw = f4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int c = 0; c < K1; ++c) {
          transpose_write(priv, G[c][0], p, q);
          transpose_write(priv + TB_FLOATS, ablk[c], p, q);
          const f4 zt = transpose_read(priv, p, q);
          const f4 at = transpose_read(priv + TB_FLOATS, p, q);
          if (c == 0) bsum = (zt[0] + zt[1]) + (zt[2] + zt[3]);
#pragma unroll
          for (int s = 0; s < 4; ++s) dw = mfma4(zt[s], at[s], dw);
        }
        f4* dst = reinterpret_cast<f4*>(lacc + w_off_p<WP>(L) + ((0 * NTH + w) * 64 + lane) * 4);
        *dst = *dst + dw;
        if (w == 0) {
          bsum += __shfl_xor(bsum, 16, 64);
          bsum += __shfl_xor(bsum, 32, 64);
          if (q == 0) lacc[P.PW + b_off_p<WP>(L) + p] += bsum;
        }
      }
      f4 z[K1];
      {
        const f4 wtl = *reinterpret_cast<const f4*>(WTp_ + w_off_p<WP>(L) + (16 * w + p) * 16 + 4 * q);
        f4 g[K1];
#pragma unroll
        for (int c = 0; c < K1; ++c) g[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < K1; ++c) g[c] = mfma4(wtl[r], G[c][0][r], g[c]);
        coop_adjoint<ACT, K1>(g, ablk, z);
      }
      __syncthreads();   // private pads (inside XZ) are done with
      // ---- hidden layers l = L-1 .. 1: state z = zbar_l (own block), ai = a_l (own block), wt = W_l^T rows 16w.. ----
      for (int l = L - 1; l >= 1; --l) {
        coop_write_blocks<K1>(XZ, w, z, p, q);
        coop_write_blocks<K1>(XA, w, ai, p, q);
        __syncthreads();
        f4 zf[K1][NTH];
        coop_read_full<K1>(XZ, zf, p, q);
        f4 g2[K1];
#pragma unroll
        for (int c = 0; c < K1; ++c) g2[c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NTH; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < K1; ++c) g2[c] = mfma4(wt[kt][r], zf[c][kt][r], g2[c]);
        // next iteration's operands
        f4 an[K1];
        f4 wtn[NTH];
        if (l >= 2) {
#pragma unroll
          for (int c = 0; c < K1; ++c) an[c] = *reinterpret_cast<const f4*>(scr + (l - 2) * 4 * SLOTB + c * 256 + lane * 4);
          load_wblk<NTH>(WTp_ + w_off_p<WP>(l - 1), w, wtn, p, q);
        }
        // dW_l rows 16w.. (all four column blocks), db_l rows 16w..
        f4 dw[NTH];
#pragma unroll
        for (int NT = 0; NT < NTH; ++NT) dw[NT] = f4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int c = 0; c < K1; ++c) {
          const f4 zt = transpose_read(XZ + (c * 4 + w) * TB_FLOATS, p, q);
          f4 at[NTH];
#pragma unroll
          for (int NT = 0; NT < NTH; ++NT) at[NT] = transpose_read(XA + (c * 4 + NT) * TB_FLOATS, p, q);
          if (c == 0) bsum = (zt[0] + zt[1]) + (zt[2] + zt[3]);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int NT = 0; NT < NTH; ++NT) dw[NT] = mfma4(zt[s], at[NT][s], dw[NT]);
        }
#pragma unroll
        for (int NT = 0; NT < NTH; ++NT) {
          f4* dst = reinterpret_cast<f4*>(lacc + w_off_p<WP>(l) + ((w * NTH + NT) * 64 + lane) * 4);
          *dst = *dst + dw[NT];
        }
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (q == 0) lacc[P.PW + b_off_p<WP>(l) + 16 * w + p] += bsum;
        coop_adjoint<ACT, K1>(g2, ai, z);       // zbar_{l-1}, own block
        if (l >= 2) {
#pragma unroll
          for (int c = 0; c < K1; ++c) ai[c] = an[c];
#pragma unroll
          for (int kt = 0; kt < NTH; ++kt) wt[kt] = wtn[kt];
        }
        __syncthreads();   // everyone has read XZ / XA
      }
      // ---- layer 0: dW_0 rows 16w.. = zbar_0 (x) input jet ---------------------------------------------
      {
        f4 b1[K1][1];
        input_jet(b1);
        f4 dw = f4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int c = 0; c < K1; ++c) {
          transpose_write(priv, z[c], p, q);
          transpose_write(priv + TB_FLOATS, b1[c][0], p, q);
          const f4 zt = transpose_read(priv, p, q);
          const f4 at = transpose_read(priv + TB_FLOATS, p, q);
          if (c == 0) bsum = (zt[0] + zt[1]) + (zt[2] + zt[3]);
#pragma unroll
          for (int s = 0; s < 4; ++s) dw = mfma4(zt[s], at[s], dw);
        }
        f4* dst = reinterpret_cast<f4*>(lacc + ((w * 1 + 0) * 64 + lane) * 4);
        *dst = *dst + dw;
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (q == 0) lacc[P.PW + b_off_p<WP>(0) + 16 * w + p] += bsum;
      }
    }
    __syncthreads();   // XZ's private pads / exchange buffers are reused by the next tile
  }

  // ---- per-workgroup results (wave 0 holds the loss sums) --------------------------------------------
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < MAX_SUMS; ++j) {
      float v = sums[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) lsum[j] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < MAX_SUMS) P.wg_sums[(int64_t)blockIdx.x * MAX_SUMS + threadIdx.x] = lsum[threadIdx.x];
  if (GRAD) {
    float* dst = P.wg_grads + (int64_t)blockIdx.x * PP;
    for (int i = threadIdx.x; i < PP; i += COOP_THREADS) dst[i] = lacc[i];
  }
}

int launch_fused_coop(int K1, bool grad, const FusedParams& P, int grid, size_t lds_bytes, hipStream_t s);

}  // namespace pinn
