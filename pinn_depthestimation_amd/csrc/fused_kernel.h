// fused_kernel.h — the MI355X fast path: one persistent kernel that, per tile of 16
// collocation points and per wave, runs
//     forward-mode jet through every layer  (dnn.py:54-55 + physics.py:6-15)
//  -> PDE residual + its adjoint             (physics.py:18-120, train.py:131-157)
//  -> reverse sweep: d loss / d theta        (train.py:191)
// entirely on v_mfma_f32_16x16x4_f32 (exact fp32, fmaf-chain numerics).
//
// Layout ("acc layout"): a 16(feature) x 16(point) tile lives in one f4 per lane;
// lane = (p = lane&15 point, q = lane>>4), register r  <->  feature 4q + r, point p.
// That is the C/D layout of the 16x16x4 MFMA, and — with the K index permuted so that
// k-step (tile kt, reg r) carries features {16kt + 4q + r : q = 0..3} — it is ALSO its B
// operand layout: a layer's accumulators feed the next layer's MFMAs with no data
// movement.  The matching A operand (weights) is then W[16MT + m][16kt + 4kq + r],
// r = 0..3 contiguous: one 16-byte load from the row-major (padded) weights.  The
// reverse sweep uses the same chain on W^T.  Only the weight-gradient GEMM
// (dW = Zbar . A^T, contraction over points) needs points on the K axis: each
// 16x16 block is transposed through a wave-private, XOR-swizzled 1 KB LDS pad.
//
// Activations needed by the reverse sweep are spilled to a per-wave global scratch
// slot in fragment-native order (fully coalesced 16 B/lane both ways; written once,
// read once).  dW/db are accumulated in a per-workgroup LDS copy of the (padded)
// gradient with plain vector read-modify-write under a per-layer wave-level lock
// (LDS fp32 atomics are ~150 cycles per wave-instruction on gfx950), written out once
// per workgroup and summed across workgroups by a second kernel in a fixed order.
#pragma once
#include <string.h>
#include <type_traits>
#include "common.h"
#include "residuals.h"

namespace pinn {

typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef PINN_FUSED_WAVES
#define PINN_FUSED_WAVES 4
#endif
#ifndef PINN_FUSED_BATCH_FLUSH
#define PINN_FUSED_BATCH_FLUSH 4   // row blocks of the LDS gradient flush read per round trip (0 = one block at a time)
#endif
#ifndef PINN_FUSED_ADJ_IN_FLUSH
#define PINN_FUSED_ADJ_IN_FLUSH 0   // 1: run the activation adjoint between the LDS flush's reads and its adds
#endif
#ifndef PINN_FUSED_MID_IO
#define PINN_FUSED_MID_IO 1   // spill stores / activation reloads issued from inside the GEMMs (see gemm_stream)
#endif
#ifndef PINN_FUSED_W16_WAVES
#define PINN_FUSED_W16_WAVES 3   // waves per SIMD the width-16 kernels are compiled for (workgroups per CU follow in pinn_fused.hip)
#endif
#ifndef PINN_FUSED_EARLY_LOCK
#define PINN_FUSED_EARLY_LOCK 1   // global gradient copy: lock + request the current values before the dW MFMAs
#endif
#ifndef PINN_FUSED_XPREF
#define PINN_FUSED_XPREF 1   // 1: request the next tile's input coordinates one tile ahead
#endif
constexpr int FUSED_WAVES = PINN_FUSED_WAVES;
constexpr int FUSED_THREADS = FUSED_WAVES * 64;
constexpr int TB_FLOATS = 256;                // one 16x16 fp32 block, XOR-swizzled (see transpose_write)
constexpr int TB_PER_WAVE = 8;                // 4 pads for the Zbar tiles + 4 for the A tiles of one quantity
constexpr int MAX_SUMS = 12;               // residual terms at [0,4), fidelity columns at [4,12)
constexpr int MSE_SUM0 = 4;

struct FusedParams {
  int d_in, d_out, L, act;
  int dir_col[PINN_MAX_DIRS];
  int64_t N, n_tiles;
  const float* X;
  const float* Wp;    // padded weights, row-major [out][in] per layer
  const float* WTp;   // padded transposed weights, row-major [in][out] per layer
  const float* Bp;    // padded biases
  float* scratch;     // activation spill, scratch_per_wave floats per wave
  int64_t scratch_per_wave;
  float* Y; float* dY;  // forward outputs (may be null)
  int loss_kind;        // bit 0: PDE residual, bit 1: fidelity MSE (both: one pass, train_newmethod.py:122-159)
  int residual_id;
  int out_col[PINN_MAX_ROLES];   // residual: output column of each role
  int q_of[PINN_MAX_DIRS];   // engine quantity (1 + direction index) of each residual direction role
  float thr, anchor; int xcol;
  const float* scale;        // residual: device term scales (null when no gradient wanted)
  int n_cols;                // mse: number of target columns
  int mse_col[PINN_MAX_ROLES];   // mse: output column of each target column
  const float* T;            // mse: targets (N, n_cols); split mode: (N - n_split, n_cols)
  int64_t n_split;           // < 0: every loss term on every point; >= 0: residual on points < n_split, mse on the rest
  const float* mse_scale;    // mse: device column scales
  float* wg_sums;            // [grid][MAX_SUMS]
  float* wg_grads;           // [grid][PP]: acc_lds: written once at kernel end; else the workgroups' live global copies
  int acc_lds;               // 1: the workgroup's gradient copy lives in LDS
  int PW, PB;                // padded weight / bias float counts
  int lds_acc_floats;        // floats reserved for the LDS gradient copy (0 if unused)
  int batch_T;               // k_fused_batch: tiles per wave and batch of the instance to launch
  int io1;                   // k_fused<..., IO1 = true>: inputs / outputs in k-step-major order (see k_fused)
  uint32_t drop_seed, drop_thresh;   // k_fused<..., DROP = true>: nn.Dropout in training mode (common.h: dropout_bits)
  float drop_scale, drop_keep;       // 1 / (1 - p), 1 - p
};

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int WP> __device__ __forceinline__ int w_off_p(int l) { return l == 0 ? 0 : WP * 16 + (l - 1) * WP * WP; }
template <int WP> __device__ __forceinline__ int b_off_p(int l) { return l * WP; }

// Weight fragments of one layer: w[MT][kt] = Wl[16MT + m][16kt + 4kq .. +3]  (Wl row-major, row
// stride 16*NT_IN).  Loaded one phase AHEAD of the GEMM that uses them (software prefetch: with
// one wave per SIMD nothing else hides the L2/MALL latency of these loads).
template <int NT_IN, int NT_OUT>
__device__ __forceinline__ void load_w(const float* __restrict__ Wl, f4 (&w)[NT_OUT][NT_IN], int m, int kq) {
  constexpr int LDW = 16 * NT_IN;
#pragma unroll
  for (int MT = 0; MT < NT_OUT; ++MT)
#pragma unroll
    for (int kt = 0; kt < NT_IN; ++kt)
      w[MT][kt] = *reinterpret_cast<const f4*>(Wl + (16 * MT + m) * LDW + 16 * kt + 4 * kq);
}

// acc[c][MT] += sum_k W[16MT + m][k] * bin[c][k]      (KR < 4: only the first KR k-steps of every k-tile carry data)
template <int NT_IN, int NT_OUT, int K1, int KR = 4>
__device__ __forceinline__ void gemm_chain(const f4 (&w)[NT_OUT][NT_IN], const f4 (&bin)[K1][NT_IN],
                                           f4 (&acc)[K1][NT_OUT]) {
#pragma unroll
  for (int MT = 0; MT < NT_OUT; ++MT)
#pragma unroll
    for (int kt = 0; kt < NT_IN; ++kt)
#pragma unroll
      for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int c = 0; c < K1; ++c) acc[c][MT] = mfma4(w[MT][kt][r], bin[c][kt][r], acc[c][MT]);
}


// One 16-row block of a layer's A operand: w[kt] = Wl[16MT + m][16kt + 4kq .. +3]  (row stride 16*NT_IN)
template <int NT_IN>
__device__ __forceinline__ void load_wblk(const float* __restrict__ Wl, int MT, f4 (&w)[NT_IN], int m, int kq) {
#pragma unroll
  for (int kt = 0; kt < NT_IN; ++kt)
    w[kt] = *reinterpret_cast<const f4*>(Wl + (16 * MT + m) * (16 * NT_IN) + 16 * kt + 4 * kq);
}

// acc[c][MT] += W[16MT + m][:] . bin[c]  with the weights STREAMED one 16-row block ahead of the
// MFMAs that use them: two blocks (8 f4 at width 64) are live instead of two whole layers, and
// nothing but `wa` is carried from one GEMM to the next (no loop-carried register copies).  `wa`
// holds block 0 on entry; while the last block computes, block 0 of the NEXT phase's matrix
// (`Wnext`, same row stride) is fetched into `wa`: its latency hides behind this GEMM's tail and
// the vector work between the two GEMMs.
// `mid` (optional) is called once, right after block 1's prefetch has been issued (block 0 for short
// GEMMs): the place for HBM traffic that must not sit in FRONT of a weight load in the in-order vmcnt
// queue.  A spill store / activation reload issued before the GEMM has to complete before the first
// streamed block (requested after it, consumed 2048 cycles later) can be used; issued here, the next
// load behind it is consumed two blocks (4096 cycles) later and its own consumer a GEMM later.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <int NT_IN, int NT_OUT, int K1, class Mid = NoMid>
__device__ __forceinline__ void gemm_stream(const float* __restrict__ Wl, const float* __restrict__ Wnext,
                                            f4 (&wa)[NT_IN], const f4 (&bin)[K1][NT_IN], f4 (&acc)[K1][NT_OUT],
                                            int m, int kq, const Mid& mid = Mid()) {
  constexpr int MID_BLOCK = NT_OUT >= 3 ? 1 : 0;
#pragma unroll
  for (int MT = 0; MT < NT_OUT; ++MT) {
    f4 wb[NT_IN];
    if (MT + 1 < NT_OUT) load_wblk<NT_IN>(Wl, MT + 1, wb, m, kq);
    else load_wblk<NT_IN>(Wnext, 0, wb, m, kq);
    if (MT == MID_BLOCK) mid();
#pragma unroll
    for (int kt = 0; kt < NT_IN; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < K1; ++c) acc[c][MT] = mfma4(wa[kt][r], bin[c][kt][r], acc[c][MT]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < NT_IN; ++kt) wa[kt] = wb[kt];
  }
}

template <int NT, int K1>
__device__ __forceinline__ void zero_tiles(f4 (&v)[K1][NT]) {
#pragma unroll
  for (int c = 0; c < K1; ++c)
#pragma unroll
    for (int MT = 0; MT < NT; ++MT) v[c][MT] = f4{0.f, 0.f, 0.f, 0.f};
}
template <int NT, int K1>
__device__ __forceinline__ void copy_tiles(f4 (&d)[K1][NT], const f4 (&srcv)[K1][NT]) {
#pragma unroll
  for (int c = 0; c < K1; ++c)
#pragma unroll
    for (int MT = 0; MT < NT; ++MT) d[c][MT] = srcv[c][MT];
}

template <int NT, int K1>
__device__ __forceinline__ void init_bias(const float* __restrict__ b, f4 (&acc)[K1][NT], int q) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT) {
    acc[0][MT] = *reinterpret_cast<const f4*>(b + 16 * MT + 4 * q);
#pragma unroll
    for (int c = 1; c < K1; ++c) acc[c][MT] = f4{0.f, 0.f, 0.f, 0.f};
  }
}

// activation and its first derivative (dnn.py:18-21): tanh ('xavier') or LeakyReLU(0.01) ('kaiming')
template <int ACT, int NT, int K1>
__device__ __forceinline__ void activate(f4 (&acc)[K1][NT]) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = acc[0][MT][r];
      float a, s;
      if constexpr (ACT == PINN_ACT_TANH) { a = tanh_f32(z); s = fmaf(-a, a, 1.f); }
      else { a = z > 0.f ? z : 0.01f * z; s = z > 0.f ? 1.f : 0.01f; }
      acc[0][MT][r] = a;
#pragma unroll
      for (int c = 1; c < K1; ++c) acc[c][MT][r] *= s;
    }
}

// adjoint of activate(): G holds (abar', abardot'_j) on entry, (zbar, zbardot_j) on exit
template <int ACT, int NT, int K1>
__device__ __forceinline__ void activate_adjoint(f4 (&G)[K1][NT], const f4 (&A)[K1][NT]) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = A[0][MT][r];
      if constexpr (ACT == PINN_ACT_TANH) {
        const float s = fmaf(-a, a, 1.f);
        float cross = 0.f;
#pragma unroll
        for (int c = 1; c < K1; ++c) {
          cross = fmaf(G[c][MT][r], A[c][MT][r], cross);
          G[c][MT][r] *= s;
        }
        G[0][MT][r] = fmaf(-2.f * a, cross, s * G[0][MT][r]);   // tanh'' = -2 a (1 - a^2)
      } else {
        const float s = a > 0.f ? 1.f : 0.01f;                   // piecewise linear: no second-derivative term
#pragma unroll
        for (int c = 0; c < K1; ++c) G[c][MT][r] *= s;
      }
    }
}


// nn.Dropout(p) after every hidden activation in training mode (dnn.py:36-38, train.py:186), fused: the keep mask is
// the counter-based hash of common.h (dropout_bits(seed, layer, unit, point)), re-derived lane-locally wherever it is
// needed — in the activation and again in its adjoint — so nothing is stored.  The point's part of the hash is formed
// once per tile (DropLane), the (layer, unit) part per register.
struct DropLane {
  uint32_t h0, hi, thresh;
  float scale, keep_p;
  __device__ __forceinline__ void init(uint32_t seed, uint32_t thr, float sc, float kp, int64_t point) {
    h0 = dropout_fmix(seed ^ ((uint32_t)point * 0x9E3779B1u));
    hi = (uint32_t)((uint64_t)point >> 32);
    thresh = thr; scale = sc; keep_p = kp;
  }
  __device__ __forceinline__ float mask(int layer, int unit) const {     // 1 / (1 - p) for a kept unit, 0 for a dropped one
    const uint32_t x = h0 ^ (((uint32_t)layer * 0x01000193u + (uint32_t)unit) * 0x9E3779B1u + hi);
    return dropout_fmix(x) >= thresh ? scale : 0.f;
  }
};

// out-of-place forms: read the accumulators, write the next GEMM's B operand / the layer adjoint
// The bias is added here rather than used as the accumulators' initial value: its load is issued
// before the GEMM and consumed after it (as an initial value its L2 latency sat exposed in front of
// every layer's first MFMA), and every accumulator chain starts from the inline constant 0.
template <int NT>
__device__ __forceinline__ void load_bias(const float* __restrict__ b, f4 (&bias)[NT], int q) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT) bias[MT] = *reinterpret_cast<const f4*>(b + 16 * MT + 4 * q);
}
template <int ACT, int NT, int K1, bool DROP = false>
__device__ __forceinline__ void activate_to(const f4 (&acc)[K1][NT], const f4 (&bias)[NT], f4 (&a)[K1][NT],
                                            const DropLane* dl = nullptr, int layer = 0, int q = 0) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = acc[0][MT][r] + bias[MT][r];
      float av, s;
      if constexpr (ACT == PINN_ACT_TANH) { av = tanh_f32(z); s = fmaf(-av, av, 1.f); }
      else { av = z > 0.f ? z : 0.01f * z; s = z > 0.f ? 1.f : 0.01f; }
      if constexpr (DROP) {      // a <- m a / (1 - p): the same mask multiplies the value and its tangents
        const float m = dl->mask(layer, 16 * MT + 4 * q + r);
        av *= m; s *= m;
      }
      a[0][MT][r] = av;
#pragma unroll
      for (int c = 1; c < K1; ++c) a[c][MT][r] = acc[c][MT][r] * s;
    }
}
// DROP: the stored jet is the masked one (a_out = m t, adot_out = m t' zdot with m = mask / (1 - p)); then
// d a_out / dz = m t', d adot_out / dz = -2 t adot_out and d adot_out / dzdot = m t': the formulas below with
// a := t = a_out (1 - p) and s := m t' (pinn_generic.hip, k_bwd_layer).  A dropped unit has a_out = adot_out = 0, s = 0.
template <int ACT, int NT, int K1, bool DROP = false>
__device__ __forceinline__ void activate_adjoint_to(const f4 (&G)[K1][NT], const f4 (&A)[K1][NT], f4 (&Z)[K1][NT],
                                                    const DropLane* dl = nullptr, int layer = 0, int q = 0) {
#pragma unroll
  for (int MT = 0; MT < NT; ++MT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = A[0][MT][r];
      if constexpr (DROP) a *= dl->keep_p;
      if constexpr (ACT == PINN_ACT_TANH) {
        float s = fmaf(-a, a, 1.f);
        if constexpr (DROP) s *= dl->mask(layer, 16 * MT + 4 * q + r);
        float cross = 0.f;
#pragma unroll
        for (int c = 1; c < K1; ++c) {
          cross = fmaf(G[c][MT][r], A[c][MT][r], cross);
          Z[c][MT][r] = G[c][MT][r] * s;
        }
        Z[0][MT][r] = fmaf(-2.f * a, cross, s * G[0][MT][r]);
      } else {
        const float s = a > 0.f ? 1.f : 0.01f;
#pragma unroll
        for (int c = 0; c < K1; ++c) Z[c][MT][r] = G[c][MT][r] * s;
      }
    }
}

template <int NT, int K1>
__device__ __forceinline__ void spill(float* __restrict__ slot, const f4 (&v)[K1][NT], int lane) {
#pragma unroll
  for (int c = 0; c < K1; ++c)
#pragma unroll
    for (int MT = 0; MT < NT; ++MT) *reinterpret_cast<f4*>(slot + (c * NT + MT) * 256 + lane * 4) = v[c][MT];
}
template <int NT, int K1>
__device__ __forceinline__ void unspill(const float* __restrict__ slot, f4 (&v)[K1][NT], int lane) {
#pragma unroll
  for (int c = 0; c < K1; ++c)
#pragma unroll
    for (int MT = 0; MT < NT; ++MT) v[c][MT] = *reinterpret_cast<const f4*>(slot + (c * NT + MT) * 256 + lane * 4);
}

// acc-layout 16x16 block (features x points) -> operand layout of the weight-gradient GEMM:
// lane (m = lane&15 feature, kq = lane>>4), element s  <-  value(feature m, point 4s + kq).
// Pad layout: row = point (64 B), 16-byte chunk c of a row stored at chunk c ^ (point & 3):
// the b32 column reads are conflict-free, the b128 row writes 2-way.  All writes of a batch are
// issued before any read so the LDS round trip is paid once per batch, not once per block
// (one wave's LDS instructions execute in program order: no barrier needed).
__device__ __forceinline__ void transpose_write(float* __restrict__ tb, f4 v, int p, int q) {
  *reinterpret_cast<f4*>(tb + p * 16 + 4 * (q ^ (p & 3))) = v;
}
template <bool CONSECUTIVE = false>   // false: element s <-> point 4s + q ; true: element s <-> point 4q + s
__device__ __forceinline__ f4 transpose_read(const float* __restrict__ tb, int p, int q) {
  f4 o;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int pt = CONSECUTIVE ? 4 * q + s : 4 * s + q;     // point index of element s for lane group q
    o[s] = tb[pt * 16 + 4 * ((p >> 2) ^ (pt & 3)) + (p & 3)];   // feature p of that point
  }
  return o;
}

// Second pad layout (PINN_FUSED_TR2, weight_grad of the tile kernel only): row = FEATURE (16 points = 64 B), the row's four
// 16-byte chunks rotated by 2 * (feature >> 2).  The write side does the transposition — lane (point p, q) scatters its
// features 4q + r as dwords (two ds_write2_b32, 2-way on the 32 write banks: free) — and the read side takes element s
// <-> point 4 kq + s (the contraction order over points is free as long as both operands use the same one): ONE
// conflict-free ds_read_b128 per block instead of two ds_read2st64_b32 behind a 2-way-conflicted ds_write_b128.
// Measured A/B (round 3, same box): 8x64 6.295 -> 6.258 ms, 10x10 on the tile kernel 0.774 -> 0.764, 100x20 24.19 -> 24.02.
#ifndef PINN_FUSED_TR2
#define PINN_FUSED_TR2 1
#endif
__device__ __forceinline__ void transpose_write2(float* __restrict__ tb, f4 v, int p, int q) {
  float* d = tb + (4 * q) * 16 + 4 * (((p >> 2) + 2 * q) & 3) + (p & 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[r * 16] = v[r];
}
__device__ __forceinline__ f4 transpose_read2(const float* __restrict__ tb, int p, int q) {   // p: feature, q: point group
  return *reinterpret_cast<const f4*>(tb + p * 16 + 4 * ((q + 2 * (p >> 2)) & 3));
}

constexpr int MAX_LOCKS = 128;

// Where dW/db contributions go.
//  LDSACC: the workgroup's LDS copy of the padded gradient.  LDS fp32 atomics (ds_add_f32)
//    measured ~150 cycles per wave-instruction on gfx950 and dominated the kernel; a plain
//    ds_read_b128 / v_add / ds_write_b128 is ~free.  So each layer's block is guarded by a
//    wave-level spin lock (one ds_cmpst by lane 0) and updated with plain vector RMW.
//  !LDSACC (gradient too large for LDS, e.g. the reference's 100x20 net): the workgroup's copy lives
//    in global memory instead and is updated the same way, under the same LDS locks — plain vector
//    read-modify-write; the waves of a workgroup share one CU's L1, so workgroup-scope fences are all
//    the coherence it takes.  (global_atomic_add_f32 into 16 shared copies was 78 % of the 100x20 step:
//    89.7 ms with, 20.0 ms without the atomics.)
// Weight blocks are stored fragment-native: tile (MT, NT), lane, reg r holds
// dW[16MT + 4(lane>>4) + r][16NT + (lane&15)]  at  woff + ((MT*NT_N + NT)*64 + lane)*4 + r.
template <bool LDSACC>
struct GradSink {
  static constexpr bool LDS = LDSACC;
  float* acc;
  int* locks;
  __device__ __forceinline__ void lock(int l, int lane) const {
    if (lane == 0) {
      int expected = 0;
      while (!__hip_atomic_compare_exchange_strong(locks + l, &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP)) {
        expected = 0;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void unlock(int l, int lane) const {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(locks + l, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void add1(int idx, float v) const { acc[idx] += v; }
  __device__ __forceinline__ void add4(int idx, f4 v) const {   // idx: float index, multiple of 4
    f4* ptr = reinterpret_cast<f4*>(acc + idx);
    *ptr = *ptr + v;
  }
};

// dW[16MT + 4q + r][16NT + n] += sum_c sum_points Z[c][MT](feature, point) * A[c][NT](feature, point)
// db[16MT + m]                += sum_points Z[0][MT](feature m, point)
// A: the layer-input jet in acc layout (registers).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `between` (optional) runs after the flush's LDS reads have been issued and before their adds: vector
// work that does not depend on the flush (the activation adjoint) then covers the LDS round trip.
template <int MT_N, int NT_N, int K1, class Sink, class Hook = NoHook>
__device__ __forceinline__ void weight_grad(const Sink& sink, int layer, int woff, int boff, const f4 (&Z)[K1][MT_N],
                                            const f4 (&A)[K1][NT_N], float* __restrict__ tb, int lane,
                                            const Hook& between = Hook()) {
  const int p = lane & 15, q = lane >> 4;
  f4 dw[MT_N][NT_N];
  float bs[MT_N];
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) dw[MT][NT] = f4{0.f, 0.f, 0.f, 0.f};
  // Global gradient copy (!Sink::LDS): take the layer's lock and request the current values NOW, so
  // that the global round trip hides behind the transposes and MFMAs below instead of sitting in the
  // flush (waves of a workgroup are on different layers almost always: holding the lock longer is free).
  f4 early[Sink::LDS ? 1 : MT_N][Sink::LDS ? 1 : NT_N];
  if constexpr (!Sink::LDS && PINN_FUSED_EARLY_LOCK) {
    sink.lock(layer, lane);
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT)
        early[MT][NT] = *reinterpret_cast<const f4*>(sink.acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4);
  }
  // Transposes run one quantity AHEAD of the MFMAs that consume them: the LDS round trip of
  // quantity c+1 (same pads: quantity c's reads have already landed in registers) overlaps the
  // MT_N*NT_N*4 MFMAs of quantity c.
  f4 zt[2][MT_N], at[2][NT_N];
  auto stage = [&](int c, f4 (&z)[MT_N], f4 (&a)[NT_N]) {
#if PINN_FUSED_TR2
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) transpose_write2(tb + MT * TB_FLOATS, Z[c][MT], p, q);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) transpose_write2(tb + (4 + NT) * TB_FLOATS, A[c][NT], p, q);
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) z[MT] = transpose_read2(tb + MT * TB_FLOATS, p, q);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) a[NT] = transpose_read2(tb + (4 + NT) * TB_FLOATS, p, q);
#else
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) transpose_write(tb + MT * TB_FLOATS, Z[c][MT], p, q);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) transpose_write(tb + (4 + NT) * TB_FLOATS, A[c][NT], p, q);
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) z[MT] = transpose_read(tb + MT * TB_FLOATS, p, q);
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) a[NT] = transpose_read(tb + (4 + NT) * TB_FLOATS, p, q);
#endif
  };
  stage(0, zt[0], at[0]);
#pragma unroll
  for (int c = 0; c < K1; ++c) {
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < K1) stage(c + 1, zt[(c + 1) & 1], at[(c + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    if (c == 0) {   // bias: zt[MT][s] = zbar(feature p, point 4s + q): sum the 4 regs here ...
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT) bs[MT] = (zt[0][MT][0] + zt[0][MT][1]) + (zt[0][MT][2] + zt[0][MT][3]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
        for (int NT = 0; NT < NT_N; ++NT) dw[MT][NT] = mfma4(zt[c & 1][MT][s], at[c & 1][NT][s], dw[MT][NT]);
  }
  // ... and the 4 lane groups here, after the last MFMA block: lgkmcnt retires in order, so a ds_bpermute
  // issued right behind the next quantity's transposes would have made quantity 0's MFMAs wait for them
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT) {
    float t = bs[MT];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    bs[MT] = t;
  }
  if constexpr (!Sink::LDS && PINN_FUSED_EARLY_LOCK) {
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT)
        *reinterpret_cast<f4*>(sink.acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4) = early[MT][NT] + dw[MT][NT];
  } else {
  sink.lock(layer, lane);
  if constexpr (PINN_FUSED_BATCH_FLUSH != 0) {   // (LDS copy or global copy alike)
    // one LDS round trip per FLUSH_ROWS row blocks (all reads issued, then adds + writes) instead of
    // one per 16x16 block: with a single wave per SIMD the serialized read-add-write chain is exposed
    constexpr int FR = PINN_FUSED_BATCH_FLUSH < MT_N ? PINN_FUSED_BATCH_FLUSH : MT_N;
#pragma unroll
    for (int M0 = 0; M0 < MT_N; M0 += FR) {
      f4 cur[FR][NT_N];
#pragma unroll
      for (int MT = M0; MT < M0 + FR && MT < MT_N; ++MT)
#pragma unroll
        for (int NT = 0; NT < NT_N; ++NT)
          cur[MT - M0][NT] = *reinterpret_cast<const f4*>(sink.acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4);
      __builtin_amdgcn_sched_barrier(0);
      if (M0 == 0) between();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int MT = M0; MT < M0 + FR && MT < MT_N; ++MT)
#pragma unroll
        for (int NT = 0; NT < NT_N; ++NT)
          *reinterpret_cast<f4*>(sink.acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4) = cur[MT - M0][NT] + dw[MT][NT];
    }
  } else {
    between();
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT) sink.add4(woff + ((MT * NT_N + NT) * 64 + lane) * 4, dw[MT][NT]);
  }
  }
  if (q == 0) {
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT) sink.add1(boff + 16 * MT + p, bs[MT]);
  }
  sink.unlock(layer, lane);
}

__device__ __forceinline__ float pick4(f4 v, int i) {
  return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
}
// output column o of the (single) output tile, for this lane's point
__device__ __forceinline__ float gather_out(f4 tile, int o, int p) {
  return __shfl(pick4(tile, o & 3), p + 16 * (o >> 2), 64);
}
template <int K1>
__device__ __forceinline__ f4 pick_q(const f4 (&out)[K1][1], int qi) {
  f4 v = out[0][0];
#pragma unroll
  for (int c = 1; c < K1; ++c) v = (qi == c) ? out[c][0] : v;
  return v;
}

// Per-lane lookup built once per kernel: which role (residual output role / mse target column)
// lands in this lane's accumulator register r2 (-1: none), and which residual quantity maps to
// engine quantity ce (-1: none).  Kept as small integers so that nothing but these is hoisted
// out of the tile loop (hoisted lane predicates would each pin an SGPR pair).
template <int K1>
struct ScatterMap {
  int role_of[4];
  int cinv[K1];
};

// roles' adjoints g[c][r] (identical in all four lane groups) -> adjoint tile G in acc layout,
// through the wave-private LDS pad (idle at this point): row (c*NR + r) holds the 16 points.
template <int K1, int NC, int NR, bool ACCUM = false>
__device__ __forceinline__ void scatter_adjoint(float* __restrict__ tb, const float (&g)[NC][NR],
                                                const ScatterMap<K1>& sm, f4 (&G)[K1][1], bool valid, int p, int q) {
  if (q == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < NR; ++r) tb[(c * NR + r) * 16 + p] = g[c][r];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ce = 0; ce < K1; ++ce)
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
      const int ci = sm.cinv[ce], ro = sm.role_of[r2];
      const bool ok = valid && ci >= 0 && ci < NC && ro >= 0;
      const float val = tb[ok ? (ci * NR + ro) * 16 + p : p];
      G[ce][0][r2] = (ACCUM ? G[ce][0][r2] : 0.f) + (ok ? val : 0.f);
    }
  __builtin_amdgcn_wave_barrier();
}

// Evaluate one residual family on the gathered jet; writes the adjoint tile G (acc layout).
template <class RES, int K1, bool GRAD>
__device__ __forceinline__ void residual_tile(const FusedParams& P, const f4 (&out)[K1][1], f4 (&G)[K1][1],
                                              float (&sums)[MAX_SUMS], const ScatterMap<K1>& sm,
                                              float* __restrict__ tb, bool valid, bool masked, int p, int q,
                                              bool primary = true) {
  constexpr int NR = RES::NR, ND = RES::ND, NT = RES::NT;
  float v[1 + ND][NR], g[1 + ND][NR], sq[NT], sc[NT];
#pragma unroll
  for (int c = 0; c <= ND; ++c) {
    const f4 tile = (c == 0) ? out[0][0] : pick_q<K1>(out, P.q_of[c - 1]);
#pragma unroll
    for (int r = 0; r < NR; ++r) v[c][r] = gather_out(tile, P.out_col[r], p);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) sc[t] = GRAD ? P.scale[t] : 0.f;
  if constexpr (std::is_same<RES, ResContinuity>::value)
    RES::template eval<GRAD>(v, sc, g, sq, P.residual_id == PINN_RES_CONTINUITY_ONLY, masked, P.anchor);
  else
    RES::template eval<GRAD>(v, sc, g, sq);
  if (valid && q == 0 && primary) {
#pragma unroll
    for (int t = 0; t < NT; ++t) sums[t] += sq[t];
  }
  if constexpr (GRAD) scatter_adjoint<K1, 1 + ND, NR>(tb, g, sm, G, valid, p, q);
}

// P.out_col entries of roles the residual does not use must be -1 (pinn_abi.hip check_spec normalises every spec):
// the search below covers all PINN_MAX_ROLES entries, and a stale 0 would claim output column 0.
template <int K1>
__device__ __forceinline__ void build_scatter_maps(const FusedParams& P, int q, ScatterMap<K1>& sm, ScatterMap<K1>& sm_mse) {
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
  sm.cinv[0] = 0; sm_mse.cinv[0] = 0;
#pragma unroll
  for (int ce = 1; ce < K1; ++ce) {
    int ci = -1;
    for (int d = PINN_MAX_DIRS - 1; d >= 0; --d) ci = (P.q_of[d] == ce) ? 1 + d : ci;
    sm.cinv[ce] = ci;
    sm_mse.cinv[ce] = -1;
  }
}

// Everything that happens on the output tile of one 16-point tile: optional Y/dY stores, PDE
// residual and/or fidelity MSE (train.py:131-157), loss partial sums, output adjoint G.
constexpr int EPI_GENERIC = 0, EPI_NS = 1, EPI_PE = 2, EPI_CONT = 3;

template <int K1, bool GRAD, bool SPLIT, int EPI = EPI_GENERIC>
__device__ __forceinline__ void loss_epilogue_impl(const FusedParams& P, const f4 (&out)[K1][1], f4 (&G)[K1][1],
                                              float (&sums)[MAX_SUMS], const ScatterMap<K1>& sm,
                                              const ScatterMap<K1>& sm_mse, float* __restrict__ tb, int64_t pt,
                                              int64_t ptc, bool valid, int p, int q, bool primary = true) {
  // EPI != 0: an epilogue specialised to ONE residual family, residual loss only, no output stores.
  // The generic epilogue keeps every family, the fidelity columns and the Y/dY stores behind runtime
  // switches; at width 64 that costs 146 spilled SGPRs (372 v_readlane per tile) against 4 with the
  // specialised form, and the 2^20-point Navier-Stokes step 1.1 % (6.89 -> 6.82 ms, same box).
  if constexpr (EPI != EPI_GENERIC) {
#pragma unroll
    for (int c = 0; c < K1; ++c) G[c][0] = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_NS) {
      if constexpr (K1 >= 4) residual_tile<ResNavierStokes, K1, GRAD>(P, out, G, sums, sm, tb, valid, false, p, q, primary);
    } else if constexpr (EPI == EPI_PE) {
      if constexpr (K1 >= 3) residual_tile<ResPhysicsEquation, K1, GRAD>(P, out, G, sums, sm, tb, valid, false, p, q, primary);
    } else {
      if constexpr (K1 >= 3) {
        const bool masked = P.residual_id == PINN_RES_CONTINUITY_ONLY && P.X[ptc * P.d_in + P.xcol] < P.thr;
        residual_tile<ResContinuity, K1, GRAD>(P, out, G, sums, sm, tb, valid, masked, p, q, primary);
      }
    }
    return;
  }
  // primary == false (cooperative kernel, waves 1..3): compute the output adjoint only — no stores, no sums
  if (P.Y != nullptr && valid && primary) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      if (o < P.d_out) {
        P.Y[pt * P.d_out + o] = out[0][0][r];
        if (P.dY != nullptr) {
#pragma unroll
          for (int c = 1; c < K1; ++c) P.dY[((int64_t)(c - 1) * P.N + pt) * P.d_out + o] = out[c][0][r];
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < K1; ++c) G[c][0] = f4{0.f, 0.f, 0.f, 0.f};
  // split mode (train.py:131-157 in one launch): collocation points first, fidelity points after them
  // SPLIT is a template parameter so that the common (unsplit) epilogue is untouched: even two extra
  // compares in this register-starved region cost the 2^20-point step 0.5-2.5 % (measured)
  const bool valid_r = SPLIT ? (valid && pt < P.n_split) : valid;
  const bool valid_m = SPLIT ? (valid && pt >= P.n_split) : valid;
  if (P.loss_kind & 1) {
    if (P.residual_id == PINN_RES_NAVIER_STOKES) {
      if constexpr (K1 >= 4) residual_tile<ResNavierStokes, K1, GRAD>(P, out, G, sums, sm, tb, valid_r, false, p, q, primary);
    } else if (P.residual_id == PINN_RES_PHYSICS_EQUATION) {
      if constexpr (K1 >= 3) residual_tile<ResPhysicsEquation, K1, GRAD>(P, out, G, sums, sm, tb, valid_r, false, p, q, primary);
    } else {
      if constexpr (K1 >= 3) {
        const bool masked = P.residual_id == PINN_RES_CONTINUITY_ONLY && P.X[ptc * P.d_in + P.xcol] < P.thr;
        residual_tile<ResContinuity, K1, GRAD>(P, out, G, sums, sm, tb, valid_r, masked, p, q, primary);
      }
    }
  }
  if (P.loss_kind & 2) {
    float gm[1][PINN_MAX_ROLES];
#pragma unroll
    for (int j = 0; j < PINN_MAX_ROLES; ++j) {
      gm[0][j] = 0.f;
      if (j < P.n_cols) {
        const float y = gather_out(out[0][0], P.mse_col[j], p);
        const int64_t trow = SPLIT ? (valid_m ? ptc - P.n_split : 0) : ptc;
        const float d = P.T[trow * P.n_cols + j] - y;                 // train.py:141 (true - pred)
        if (valid_m && q == 0 && primary) sums[MSE_SUM0 + j] += d * d;
        if (GRAD) gm[0][j] = -2.f * P.mse_scale[j] * d;
      }
    }
    if constexpr (GRAD) scatter_adjoint<K1, 1, PINN_MAX_ROLES, true>(tb, gm, sm_mse, G, valid_m, p, q);
  }
}

// SPLIT_OK = false leaves the split-mode code out of the kernel altogether: k_fused at width 64 is so
// register-starved around the epilogue that even a never-taken second copy of it cost the 2^20-point
// step 5 % (and two extra compares in the single copy 0.5-2.5 %); the host runs split requests that
// would land on that kernel as two passes instead (pinn_fused.hip).
template <int K1, bool GRAD, bool SPLIT_OK = true, int EPI = EPI_GENERIC>
__device__ __forceinline__ void loss_epilogue(const FusedParams& P, const f4 (&out)[K1][1], f4 (&G)[K1][1],
                                              float (&sums)[MAX_SUMS], const ScatterMap<K1>& sm,
                                              const ScatterMap<K1>& sm_mse, float* __restrict__ tb, int64_t pt,
                                              int64_t ptc, bool valid, int p, int q, bool primary = true) {
  if constexpr (SPLIT_OK && EPI == EPI_GENERIC) {
    if (P.n_split >= 0) {
      loss_epilogue_impl<K1, GRAD, true>(P, out, G, sums, sm, sm_mse, tb, pt, ptc, valid, p, q, primary);
      return;
    }
  }
  loss_epilogue_impl<K1, GRAD, false, EPI>(P, out, G, sums, sm, sm_mse, tb, pt, ptc, valid, p, q, primary);
}

// Diagnostic build only (-DPINN_DIAG): s_memtime stamps per phase, printed by wave 0 of block 0.
// Stamps drain the memory counters, so read the SHARES, never the total (cdna guide §7).
#ifdef PINN_DIAG
#define PINN_STAMP(i)                                                                      \
  do {                                                                                     \
    unsigned long long t_;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    diag[i] += t_ - tprev;                                                                 \
    tprev = t_;                                                                            \
  } while (0)
#else
#define PINN_STAMP(i) do { } while (0)
#endif

// KRO > 0 (the specialised-epilogue kernels of width 64): network inputs and outputs sit in K-STEP-MAJOR order — input
// column f at padded index perm16(f) = 4 (f & 3) + (f >> 2), i.e. lane group f, register 0 for d_in <= 4, and output o at
// padded row perm16(o) — so the first layer's forward GEMM is ONE k-step instead of four and the output layer's reverse
// GEMM KRO = ceil(d_out / 4) k-steps (Navier-Stokes 3 -> 8x64 -> 4: 96 of the tile's 5696 MFMAs gone).  The packing
// kernels place the weights accordingly (pinn_fused.hip, PACK_IN / PACK_OUT) and the host hands the epilogue PADDED row
// indices in out_col / dir_col, so the epilogue code is the same.
template <int WP, int K1, bool GRAD, bool LDSACC, int ACT, int EPI = EPI_GENERIC, int KRO = 0, bool DROP = false>
__global__ __launch_bounds__(FUSED_THREADS, WP == 16 ? PINN_FUSED_W16_WAVES : FUSED_WAVES / 4) void k_fused(const FusedParams P) {
  static_assert(!DROP || (ACT == PINN_ACT_TANH && KRO == 0), "dropout instances: tanh, natural unit order");
  constexpr bool IO1 = KRO > 0;
  constexpr int KRI = IO1 ? 1 : 4;          // k-steps of the first layer's contraction (d_in <= 4 when IO1)
  constexpr int KRL = IO1 ? KRO : 4;        // k-steps of the output layer's reverse contraction
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTH = WP / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  float* lacc = smem;
  int* locks = reinterpret_cast<int*>(smem + P.lds_acc_floats);
  float* tb = smem + P.lds_acc_floats + MAX_LOCKS + wave * (TB_PER_WAVE * TB_FLOATS);
  float* lsum = smem + P.lds_acc_floats + MAX_LOCKS + FUSED_WAVES * TB_PER_WAVE * TB_FLOATS;
  const int PP = P.PW + P.PB;
  GradSink<LDSACC> sink;
  sink.acc = LDSACC ? lacc : P.wg_grads + (int64_t)blockIdx.x * PP;
  sink.locks = locks;
  if (GRAD) {
    if (LDSACC) {
      for (int i = threadIdx.x; i < PP; i += FUSED_THREADS) lacc[i] = 0.f;
    }
    if (threadIdx.x < MAX_LOCKS) locks[threadIdx.x] = 0;
    __syncthreads();
  }
  float sums[MAX_SUMS];
#pragma unroll
  for (int j = 0; j < MAX_SUMS; ++j) sums[j] = 0.f;

  ScatterMap<K1> sm, sm_mse;
  build_scatter_maps<K1>(P, q, sm, sm_mse);
  const int gw = blockIdx.x * FUSED_WAVES + wave, nw = gridDim.x * FUSED_WAVES;
  // restrict-qualified views: weights / biases / inputs are read-only for the whole launch and the
  // spill slot is private to this wave, so loads may be scheduled across the spill stores
  float* __restrict__ scr = P.scratch + (int64_t)gw * P.scratch_per_wave;
  const float* __restrict__ Wp_ = P.Wp;
  const float* __restrict__ WTp_ = P.WTp;
  const float* __restrict__ Bp_ = P.Bp;
  constexpr int SLOT = K1 * NTH * 256;  // floats per spilled layer
  const int L = P.L;

#ifdef PINN_DIAG
  unsigned long long diag[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
#if PINN_FUSED_XPREF
  auto load_x = [&](int64_t t, f4& x) {
    int64_t pc = t * 16 + p;
    pc = pc < P.N ? pc : P.N - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = IO1 ? (r == 0 ? q : 16) : 4 * q + r;      // (IO1: column q in register 0, nothing else)
      x[r] = (f < P.d_in) ? P.X[pc * P.d_in + f] : 0.f;
    }
  };
  f4 xnext;
  load_x(gw < P.n_tiles ? gw : 0, xnext);
#endif
  for (int64_t tile = gw; tile < P.n_tiles; tile += nw) {
    PINN_STAMP(11);
    const int64_t pt = tile * 16 + p;
    const bool valid = pt < P.N;
    const int64_t ptc = valid ? pt : P.N - 1;
    DropLane dl;
    if constexpr (DROP) dl.init(P.drop_seed, P.drop_thresh, P.drop_scale, P.drop_keep, pt);
#if PINN_FUSED_XPREF
    const f4 xcur = xnext;
    load_x(tile + nw < P.n_tiles ? tile + nw : tile, xnext);
#endif
    // ---- layer-0 input jet: features 4q + r of (x, unit tangents) --------------------------
    auto input_jet = [&](f4 (&b)[K1][1]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * q + r;                 // PADDED input index (dir_col holds padded indices when IO1)
#if PINN_FUSED_XPREF
        b[0][0][r] = xcur[r];
#else
        const int fr = IO1 ? (r == 0 ? q : 16) : f;
        b[0][0][r] = (fr < P.d_in) ? P.X[ptc * P.d_in + fr] : 0.f;
#endif
#pragma unroll
        for (int c = 1; c < K1; ++c) b[c][0][r] = (f == P.dir_col[c - 1]) ? 1.f : 0.f;
      }
    };
    f4 b0[K1][1];
    input_jet(b0);
    // ---- forward chain: weights streamed block-by-block (gemm_stream), activations written straight
    // into the next GEMM's B operand: nothing is copied between layers ------------------------------
    f4 a[K1][NTH];
    f4 ws[NTH];   // the weight block the next GEMM starts with
    {
      f4 w0[NTH][1];
      load_w<1, NTH>(Wp_, w0, p, q);
      load_wblk<NTH>(Wp_ + w_off_p<WP>(L > 1 ? 1 : L), 0, ws, p, q);
      f4 bias[NTH];
      load_bias<NTH>(Bp_ + b_off_p<WP>(0), bias, q);
      f4 acc0[K1][NTH];
      zero_tiles<NTH, K1>(acc0);
      gemm_chain<1, NTH, K1, KRI>(w0, b0, acc0);
      PINN_STAMP(0);
      activate_to<ACT, NTH, K1, DROP>(acc0, bias, a, &dl, 0, q);
    }
#if !PINN_FUSED_MID_IO
    if (GRAD && L > 1) spill<NTH, K1>(scr, a, lane);      // a_L itself stays in registers for the reverse sweep
#endif
    PINN_STAMP(1);
    for (int l = 1; l < L; ++l) {
      f4 bias[NTH];
      load_bias<NTH>(Bp_ + b_off_p<WP>(l), bias, q);
      f4 nx[K1][NTH];
      zero_tiles<NTH, K1>(nx);
#if PINN_FUSED_MID_IO
      // a_l (this GEMM's B operand) is spilled from INSIDE the GEMM (see gemm_stream); a_L is never spilled
      auto sp = [&]() { if (GRAD) spill<NTH, K1>(scr + (l - 1) * SLOT, a, lane); };
      gemm_stream<NTH, NTH, K1>(Wp_ + w_off_p<WP>(l), Wp_ + w_off_p<WP>(l + 1), ws, a, nx, p, q, sp);
      PINN_STAMP(0);
      activate_to<ACT, NTH, K1, DROP>(nx, bias, a, &dl, l, q);
#else
      gemm_stream<NTH, NTH, K1>(Wp_ + w_off_p<WP>(l), Wp_ + w_off_p<WP>(l + 1), ws, a, nx, p, q);
      PINN_STAMP(0);
      activate_to<ACT, NTH, K1, DROP>(nx, bias, a, &dl, l, q);
      if (GRAD && l < L - 1) spill<NTH, K1>(scr + l * SLOT, a, lane);   // (the last hidden jet is never re-read)
#endif
      PINN_STAMP(1);
    }
    f4 out[K1][1];
    {
      f4 bias_o[1];
      load_bias<1>(Bp_ + b_off_p<WP>(L), bias_o, q);
      zero_tiles<1, K1>(out);
      // (the block fetched behind the output GEMM is W_{L-1}^T's first: the reverse sweep starts there)
      gemm_stream<NTH, 1, K1>(Wp_ + w_off_p<WP>(L), WTp_ + w_off_p<WP>(L > 1 ? L - 1 : 0), ws, a, out, p, q);
      out[0][0] += bias_o[0];
    }
    // reverse-sweep operands whose latency the residual evaluation below hides
    f4 wtl[NTH][1];
    f4 ai[K1][NTH];
    if constexpr (GRAD) {
      load_w<1, NTH>(WTp_ + w_off_p<WP>(L), wtl, p, q);
#if !PINN_FUSED_MID_IO
      unspill<NTH, K1>(scr + (L > 1 ? L - 2 : 0) * SLOT, ai, lane);            // a_{L-1}
#endif
    }

    // ---- outputs / loss -----------------------------------------------------------------------
    f4 G[K1][1];
    loss_epilogue<K1, GRAD, (WP < 64), EPI>(P, out, G, sums, sm, sm_mse, tb, pt, ptc, valid, p, q);

    PINN_STAMP(2);
    // ---- reverse sweep ------------------------------------------------------------------------
    // State entering iteration l: z = zbar of hidden layer l, ai = a_l (its load in flight), ws =
    // first block of W_l^T.  The iteration runs  abar_l = W_l^T z  (a_l lands meanwhile), then
    // dW_l = z (x) a_l, then z <- adjoint(abar_l, a_l) = zbar of layer l-1, and only then re-uses
    // ai's registers for a_{l-1}: each spilled layer is read once, into the registers it is used
    // from, and there are no loop-carried copies.
    if constexpr (GRAD) {
      weight_grad<1, NTH, K1>(sink, L, w_off_p<WP>(L), P.PW + b_off_p<WP>(L), G, a, tb, lane);
      f4 z[K1][NTH];
      {
        f4 g[K1][NTH];
        zero_tiles<NTH, K1>(g);
        gemm_chain<1, NTH, K1, KRL>(wtl, G, g);
        activate_adjoint_to<ACT, NTH, K1, DROP>(g, a, z, &dl, L - 1, q);
      }
      for (int l = L - 1; l >= 1; --l) {
        PINN_STAMP(3);
        f4 g2[K1][NTH];
        zero_tiles<NTH, K1>(g2);
#if PINN_FUSED_MID_IO
        // a_l is requested from inside the GEMM (see gemm_stream) and used by the weight gradient after it
        auto ld = [&]() { unspill<NTH, K1>(scr + (l - 1) * SLOT, ai, lane); };
        gemm_stream<NTH, NTH, K1>(WTp_ + w_off_p<WP>(l), WTp_ + w_off_p<WP>(l >= 2 ? l - 1 : 1), ws, z, g2, p, q, ld);
#else
        gemm_stream<NTH, NTH, K1>(WTp_ + w_off_p<WP>(l), WTp_ + w_off_p<WP>(l >= 2 ? l - 1 : 1), ws, z, g2, p, q);
#endif
        PINN_STAMP(6);
#if PINN_FUSED_ADJ_IN_FLUSH
        f4 zn[K1][NTH];
        auto adj = [&]() { activate_adjoint_to<ACT, NTH, K1, DROP>(g2, ai, zn, &dl, l - 1, q); };
        weight_grad<NTH, NTH, K1>(sink, l, w_off_p<WP>(l), P.PW + b_off_p<WP>(l), z, ai, tb, lane, adj);
        copy_tiles<NTH, K1>(z, zn);
#if !PINN_FUSED_MID_IO
        if (l >= 2) unspill<NTH, K1>(scr + (l - 2) * SLOT, ai, lane);            // a_{l-1}
#endif
#else
        weight_grad<NTH, NTH, K1>(sink, l, w_off_p<WP>(l), P.PW + b_off_p<WP>(l), z, ai, tb, lane);
        PINN_STAMP(5);
        activate_adjoint_to<ACT, NTH, K1, DROP>(g2, ai, z, &dl, l - 1, q);
#if !PINN_FUSED_MID_IO
        if (l >= 2) unspill<NTH, K1>(scr + (l - 2) * SLOT, ai, lane);            // a_{l-1}
#endif
#endif
        PINN_STAMP(4);
      }
      {  // layer 0: z = zbar_0, input = (x, unit tangents)
        f4 b1[K1][1];
        input_jet(b1);   // recomputed rather than kept live across the whole tile
        weight_grad<NTH, 1, K1>(sink, 0, 0, P.PW + b_off_p<WP>(0), z, b1, tb, lane);
      }
    }
    PINN_STAMP(7);
  }
#ifdef PINN_DIAG
  if (blockIdx.x == 0 && threadIdx.x == 0 && GRAD) {
    unsigned long long tot = 0;
    for (int i = 0; i < 12; ++i) tot += diag[i];
    printf("DIAG cycles/wave: fwd_gemm %llu fwd_act+spill %llu out+residual %llu | unspill_wait %llu adjoint %llu wgrad(transp+mfma+flush) %llu bwd_gemm %llu first/last-layer %llu loop-top %llu | total %llu\n",
           diag[0], diag[1], diag[2], diag[3], diag[4], diag[5], diag[6], diag[7], diag[11], tot);
  }
#endif

  // ---- per-workgroup reductions ------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < MAX_SUMS; ++j) {
    float v = sums[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) lsum[wave * MAX_SUMS + j] = v;
  }
  __syncthreads();
  if (threadIdx.x < MAX_SUMS) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < FUSED_WAVES; ++w) v += lsum[w * MAX_SUMS + threadIdx.x];
    P.wg_sums[(int64_t)blockIdx.x * MAX_SUMS + threadIdx.x] = v;
  }
  if (GRAD && LDSACC) {
    float* dst = P.wg_grads + (int64_t)blockIdx.x * PP;
    for (int i = threadIdx.x; i < PP; i += FUSED_THREADS) dst[i] = lacc[i];
  }
}

// launcher implemented once per WP in pinn_fused_wXX.hip
template <int WP>
int launch_fused(int K1, bool grad, const FusedParams& P, int grid, size_t lds_bytes, hipStream_t s);

}  // namespace pinn
