// fused_batch_kernel.h — the fused chain kernel for NARROW networks at large point counts:
// the reference's own shapes, 2->10x10->6 (config_CMB.json:2-9) and 2->100x20->3 (config_CMB_h.json:2-7;
// config.json / config_txyz.json are 100x20 / 20x20 too).
//
// Why another kernel.  k_fused walks ONE 16-point tile per wave through all layers.  At hidden width 10 / 20 that
// is 12 / 48 MFMAs per layer between a weight fetch, a bias fetch, a spill store, a lock and a gradient flush,
// and rocprofv3 shows where the time goes (profiles/r03/pe10x10_*, co100x20_*): both shapes sit on HBM, not on the
// matrix pipe — 3.8 KB per point and step at 5.3-5.7 TB/s for 10x10 (the padded activation spill), 120 KB per
// point at 5.0 TB/s for 100x20 (76 KB of padded spill + 44 KB of read-modify-write on the per-workgroup gradient
// copies, one 4 KB block per tile and layer).  Three changes, all about bytes and per-layer fixed costs:
//
//  * LAYER-MAJOR BATCHES.  A wave owns T tiles (T*16 points) at a time and takes all of them through layer l
//    before layer l+1: the layer's weights / bias are fetched once per T tiles, and — the point — the weight
//    gradient of layer l is accumulated in registers over the T tiles and flushed ONCE (1/T of the lock traffic
//    and of the gradient read-modify-write).  The T jets are the wave's registers (15 per tile at width 20, k = 2).
//  * K-STEP-MAJOR FEATURE ORDER.  Hidden unit f sits at padded index 16*(f/16) + perm16(f%16), perm16(c) =
//    4*(c&3) + (c>>2): in the accumulator layout of fused_kernel.h that is k-step s = f/4, lane group f%4, so a
//    contraction over W features is ceil(W/4) k-steps (3 instead of 4 at width 10, 5 instead of 8 at width 20)
//    and the activation touches ceil(W/4) registers instead of WP/4.  A permutation of hidden units leaves the
//    network function alone; it lives entirely in the packing / un-packing kernels (pinn_fused.hip, `perm`).
//    Inputs use the same order (d_in <= 4: ONE k-step for the first layer); outputs stay in natural order so the
//    residual epilogue of fused_kernel.h is shared unchanged.
//  * COMPACT SPILL.  Only the live registers of a jet are written: K1 * ceil(W/4) dwords per lane and layer
//    (1280 B per 16-point tile, quantity and layer at width 20 instead of 2048).
//
// Everything else — accumulator layout = next operand layout, transposes through wave-private swizzled LDS pads for
// the weight gradient, the workgroup's gradient copy under per-layer locks, the loss epilogue — is fused_kernel.h's.
#pragma once
#include "fused_kernel.h"

namespace pinn {

__host__ __device__ constexpr int perm16(int c) { return 4 * (c & 3) + (c >> 2); }   // an involution on 0..15

// compile-time loop over the tiles of a batch: the tile index must be a constant at IR generation so that the T jets
// are scalarised into registers (a `#pragma unroll` loop is unrolled too late for that: the array went to scratch)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

#ifndef PINN_BATCH_T
#define PINN_BATCH_T 0     // 0: batch_tiles(); > 0 forces a value (experiments)
#endif
#ifndef PINN_BATCH_OCC
#define PINN_BATCH_OCC 0     // 0: batch_occ(); > 0 forces a value (experiments)
#endif
#ifndef PINN_BATCH_OCC16
#define PINN_BATCH_OCC16 2   // width <= 16: two waves per SIMD (256 registers each) with half the tiles per batch
#endif
#ifndef PINN_BATCH_PF
#define PINN_BATCH_PF 4       // tiles of spilled activations in flight ahead of the reverse sweep
#endif
constexpr int BATCH_PF = PINN_BATCH_PF;
#ifndef PINN_BATCH_SKIP
#define PINN_BATCH_SKIP 0     // diagnostic builds only (wrong results): 1 no spill stores, 2 no spill loads, 4 no LDS transposes,
#endif                        // 8 no weight-gradient MFMAs, 16 no tanh, 32 no gradient flush
// Tiles per wave and batch: as many as the 512-register file takes WITHOUT a spill (a scratch reload shares the
// vector-memory counter with the activation prefetch and drains it: bwg_* comment).  hipcc's resource report per
// instance: width <= 16: 8 tiles at K1 = 3, 4 at K1 = 4; width <= 32: 4 at K1 = 3, 2 at K1 = 4.
__host__ __device__ constexpr int batch_occ(int WP, int K1) {     // waves per SIMD (= workgroups per CU)
  return PINN_BATCH_OCC > 0 ? PINN_BATCH_OCC : (WP == 16 ? PINN_BATCH_OCC16 : 1);
}
__host__ __device__ constexpr int batch_tiles(int WP, int K1) {
  return PINN_BATCH_T > 0 ? PINN_BATCH_T
       : WP == 16 ? (batch_occ(WP, K1) == 2 ? (K1 <= 3 ? 4 : 2) : (K1 <= 3 ? 8 : 4)) : (K1 <= 3 ? 4 : 2);
}
constexpr int BATCH_WAVES = 4;
constexpr int BATCH_THREADS = BATCH_WAVES * 64;
__host__ __device__ constexpr int batch_pads(int WP, int K1) { return K1 * 2 * (WP / 16) > 4 ? K1 * 2 * (WP / 16) : 4; }   // 1 KB pads per wave

// A-operand fragments of one layer, all output tiles: w[MT][kt] = Wl[16MT + m][16kt + 4kq .. +3]  (row stride LDW)
template <int NKT, int NT_OUT>
__device__ __forceinline__ void bload_w(const float* __restrict__ Wl, int LDW, f4 (&w)[NT_OUT][NKT], int m, int kq) {
#pragma unroll
  for (int MT = 0; MT < NT_OUT; ++MT)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) w[MT][kt] = *reinterpret_cast<const f4*>(Wl + (16 * MT + m) * LDW + 16 * kt + 4 * kq);
}

// acc[c][MT] += sum_{s < KSIN} w[MT][s] * b[c][s]   (k-step s = 4 kt + r; issue order keeps K1*NT_OUT chains apart)
template <int KSIN, int NKT, int NT_OUT, int K1>
__device__ __forceinline__ void bgemm(const f4 (&w)[NT_OUT][NKT], const float (&b)[K1][KSIN], f4 (&acc)[K1][NT_OUT]) {
#pragma unroll
  for (int s = 0; s < KSIN; ++s)
#pragma unroll
    for (int MT = 0; MT < NT_OUT; ++MT)
#pragma unroll
      for (int c = 0; c < K1; ++c) acc[c][MT] = mfma4(w[MT][s >> 2][s & 3], b[c][s], acc[c][MT]);
}

template <int ACT, int KS, int NTH, int K1>
__device__ __forceinline__ void bactivate(const f4 (&acc)[K1][NTH], const f4 (&bias)[NTH], float (&a)[K1][KS]) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float z = acc[0][s >> 2][s & 3] + bias[s >> 2][s & 3];
    float av, sd;
    if constexpr (ACT == PINN_ACT_TANH) { av = (PINN_BATCH_SKIP & 16) ? z : tanh_f32(z); sd = fmaf(-av, av, 1.f); }
    else { av = z > 0.f ? z : 0.01f * z; sd = z > 0.f ? 1.f : 0.01f; }
    a[0][s] = av;
#pragma unroll
    for (int c = 1; c < K1; ++c) a[c][s] = acc[c][s >> 2][s & 3] * sd;
  }
}

// Z <- adjoint of the activation at jet A, applied to G (acc layout).  Z may be the same array as A.
template <int ACT, int KS, int NTH, int K1>
__device__ __forceinline__ void badjoint(const f4 (&G)[K1][NTH], const float (&A)[K1][KS], float (&Z)[K1][KS]) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float a = A[0][s];
    if constexpr (ACT == PINN_ACT_TANH) {
      const float sd = fmaf(-a, a, 1.f);
      float cross = 0.f;
      float zc[K1];
#pragma unroll
      for (int c = 1; c < K1; ++c) {
        cross = fmaf(G[c][s >> 2][s & 3], A[c][s], cross);
        zc[c] = G[c][s >> 2][s & 3] * sd;
      }
      zc[0] = fmaf(-2.f * a, cross, sd * G[0][s >> 2][s & 3]);
#pragma unroll
      for (int c = 0; c < K1; ++c) Z[c][s] = zc[c];
    } else {
      const float sd = a > 0.f ? 1.f : 0.01f;
#pragma unroll
      for (int c = 0; c < K1; ++c) Z[c][s] = G[c][s >> 2][s & 3] * sd;
    }
  }
}

// Compact spill slot of one (tile, layer): quantity c, k-steps [4kt, 4kt+4) at float offset (c*KS + 4kt)*64;
// a full group of four is one 16-byte access per lane, the ragged last group 1-3 dword accesses in [j][lane] order.
template <int KS, int K1>
__device__ __forceinline__ void bspill(float* __restrict__ slot, const float (&a)[K1][KS], int lane) {
#pragma unroll
  for (int c = 0; c < K1; ++c) {
#pragma unroll
    for (int kt = 0; 4 * kt < KS; ++kt) {
      float* d = slot + (c * KS + 4 * kt) * 64;
      if (4 * kt + 4 <= KS) {
        *reinterpret_cast<f4*>(d + lane * 4) = f4{a[c][4 * kt], a[c][4 * kt + 1], a[c][4 * kt + 2], a[c][4 * kt + 3]};
      } else {
#pragma unroll
        for (int j = 0; 4 * kt + j < KS; ++j) d[j * 64 + lane] = a[c][4 * kt + j];
      }
    }
  }
}
template <int KS, int K1>
__device__ __forceinline__ void bunspill(const float* __restrict__ slot, float (&a)[K1][KS], int lane) {
#pragma unroll
  for (int c = 0; c < K1; ++c) {
#pragma unroll
    for (int kt = 0; 4 * kt < KS; ++kt) {
      const float* d = slot + (c * KS + 4 * kt) * 64;
      if (4 * kt + 4 <= KS) {
        const f4 v = *reinterpret_cast<const f4*>(d + lane * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[c][4 * kt + r] = v[r];
      } else {
#pragma unroll
        for (int j = 0; 4 * kt + j < KS; ++j) a[c][4 * kt + j] = d[j * 64 + lane];
      }
    }
  }
}

// Weight-gradient operands of one tile.  The caller places the steps around independent work so that neither LDS
// trip is waited for (one wave per SIMD: nothing else hides it), and the helpers keep register live ranges short —
// this kernel lives at the edge of the 512-register file, and a spilled register is worse than it looks: scratch
// reloads share the vector-memory counter with the activation prefetch, so every reload drains the prefetch ring.
//   bwg_write_ks / _f4   one quantity's tiles -> wave-private swizzled pads (transpose_write); each 16-byte quad is
//                        formed right before its store.   ... the caller's next GEMM runs meanwhile ...
//   bwg_read_q           one quantity's transposed operands -> registers (16 at width 32).
//   bwg_tail             quantity c+1 is read while the MFMAs of quantity c run (double buffer); quantity 0 has been
//                        read by the caller before its VALU work.  dw[MT][NT] += sum_c sum_points Z (x) A;
//                        bs[MT] += this lane's share of sum_points Z[0][MT] (lane groups combined at flush time).
// Pads of quantity c: tb + c*(MT_N+NT_N) KB, Z tiles first.  One wave's LDS operations execute in order: no barrier.
// pad layout of the weight-gradient operands: fused_kernel.h's second one (feature-major rows, PINN_FUSED_TR2) or the first
#ifndef PINN_BATCH_TR2
#define PINN_BATCH_TR2 PINN_FUSED_TR2
#endif
__device__ __forceinline__ void btr_write(float* __restrict__ tb, f4 v, int p, int q) {
#if PINN_BATCH_TR2
  transpose_write2(tb, v, p, q);
#else
  transpose_write(tb, v, p, q);
#endif
}
__device__ __forceinline__ f4 btr_read(const float* __restrict__ tb, int p, int q) {
#if PINN_BATCH_TR2
  return transpose_read2(tb, p, q);
#else
  return transpose_read(tb, p, q);
#endif
}
template <int NT, int KS>
__device__ __forceinline__ void bwg_write_ks(float* __restrict__ tbq, const float (&v)[KS], int p, int q) {
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
    btr_write(tbq + kt * TB_FLOATS, f4{4 * kt < KS ? v[4 * kt < KS ? 4 * kt : 0] : 0.f,
                                             4 * kt + 1 < KS ? v[4 * kt + 1 < KS ? 4 * kt + 1 : 0] : 0.f,
                                             4 * kt + 2 < KS ? v[4 * kt + 2 < KS ? 4 * kt + 2 : 0] : 0.f,
                                             4 * kt + 3 < KS ? v[4 * kt + 3 < KS ? 4 * kt + 3 : 0] : 0.f}, p, q);
}
template <int MT_N, int NT_N>
__device__ __forceinline__ void bwg_read_q(f4 (&zt)[MT_N], f4 (&at)[NT_N], const float* __restrict__ tbq, int p, int q) {
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT) zt[MT] = btr_read(tbq + MT * TB_FLOATS, p, q);
#pragma unroll
  for (int NT = 0; NT < NT_N; ++NT) at[NT] = btr_read(tbq + (MT_N + NT) * TB_FLOATS, p, q);
}
// NACC accumulator sets (dw[a]): a single 16x16 block (width <= 16) would otherwise be ONE dependent MFMA chain,
// 40 cycles per link instead of the 32 of the issue rate; the sets are summed at flush time.
template <int MT_N, int NT_N, int K1, int NACC>
__device__ __forceinline__ void bwg_tail(f4 (&dw)[NACC][MT_N][NT_N], float (&bs)[MT_N], f4 (&zt)[2][MT_N], f4 (&at)[2][NT_N],
                                         const float* __restrict__ tb, int p, int q) {
#pragma unroll
  for (int c = 0; c < K1; ++c) {
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < K1) bwg_read_q<MT_N, NT_N>(zt[(c + 1) & 1], at[(c + 1) & 1], tb + (c + 1) * (MT_N + NT_N) * TB_FLOATS, p, q);
    __builtin_amdgcn_sched_barrier(0);
    if (c == 0) {
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT) bs[MT] += (zt[0][MT][0] + zt[0][MT][1]) + (zt[0][MT][2] + zt[0][MT][3]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
        for (int NT = 0; NT < NT_N; ++NT)
          dw[s % NACC][MT][NT] = mfma4(zt[c & 1][MT][s], at[c & 1][NT][s], dw[s % NACC][MT][NT]);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int MT_N, int NT_N, int NACC>
__device__ __forceinline__ void bwg_zero(f4 (&dw)[NACC][MT_N][NT_N], float (&bs)[MT_N]) {
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT) {
    bs[MT] = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT) dw[a][MT][NT] = f4{0.f, 0.f, 0.f, 0.f};
  }
}

// Where a batch's dW / db go — once per layer and batch, so neither form takes a lock:
//  BSINK_LDS_WAVE  a gradient copy per WAVE in LDS (4 x PP floats: the 10x10 net's 12 KB each), plain ds read-add-write;
//                  the four copies are summed when the workgroup writes its result out.
//  BSINK_ATOMIC    gradient too large for that (100x20: 410 KB): no-return global_atomic_add_f32 into one of a few
//                  shared copies in HBM, fire and forget (nothing is read, nothing is waited for).  The copies are
//                  stored REGISTER-MAJOR inside each 16x16 block ([r][lane] instead of [lane][r]) so that one wave
//                  instruction adds 256 contiguous bytes (the shape the memory-side atomic units take at full rate).
//                  The per-workgroup copies under LDS locks of k_fused cost this kernel 5 of 24.7 ms at 100x20: four
//                  waves in lockstep queueing for a global read-modify-write per layer.
constexpr int BSINK_LDS_WAVE = 0, BSINK_ATOMIC = 1;
constexpr int BATCH_ATOMIC_COPIES = 16;

// RKS / CKS: k-steps that carry real units on the block's rows / columns (k-step-major order: row 4q + r of tile MT is
// k-step 4 MT + r, column p of tile NT is k-step 4 NT + (p & 3)).  The atomic sink adds only those: 1.6 KB instead of
// 4 KB per layer at width 20 (the rest of the padded block is exact zeros).  Experiments of round 3 around this flush,
// all measured and dropped: (1) all lanes adding, zeros included — the instruction count becomes a compile-time constant,
// so the compiler's later waits stay counted instead of vmcnt(0) behind the exec-masked branches — costs more than it
// gains, the atomic units are the scarcer resource (2^20 points of 100x20 14.9 -> 15.9 ms, 12 514 points 375 -> 455 us);
// (2) at 12 514 points, a tile's weight gradient handed to a helper wave on the same SIMD through two LDS pad buffers
// (4 main + 4 helper waves per workgroup): 378 us against 375 — the SIMD's matrix pipe is shared and the chain is
// bound by its vector-memory waits, not by the 48 MFMAs taken off it; (3) buffer atomics with out-of-range offsets for
// the masked lanes (static count AND no extra bytes): 365 us at 12 514 points, but the 512-register instances aborted
// on the GPU box — not pursued on a shared machine.
template <int MT_N, int NT_N, int NACC, int SINK, int RKS = 4 * MT_N, int CKS = 4 * NT_N>
__device__ __forceinline__ void bwgrad_flush(float* __restrict__ acc, int woff, int boff, const f4 (&dwa)[NACC][MT_N][NT_N],
                                             float (&bs)[MT_N], int lane) {
  const int p = lane & 15, q = lane >> 4;
  f4 dw[MT_N][NT_N];
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) {
      dw[MT][NT] = dwa[0][MT][NT];
#pragma unroll
      for (int a = 1; a < NACC; ++a) dw[MT][NT] += dwa[a][MT][NT];
    }
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT) {
    float t = bs[MT];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    bs[MT] = t;
  }
  if constexpr (SINK == BSINK_ATOMIC) {
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * MT + r >= RKS) continue;                                  // (compile time: a row k-step of zeros)
          if (4 * NT + 4 <= CKS || 4 * NT + (p & 3) < CKS)                  // lanes whose column carries a real unit
            __hip_atomic_fetch_add(acc + woff + ((MT * NT_N + NT) * 4 + r) * 64 + lane, dw[MT][NT][r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    if (q == 0) {
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT)
        __hip_atomic_fetch_add(acc + boff + 16 * MT + p, bs[MT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    f4 cur[MT_N][NT_N];
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT)
        cur[MT][NT] = *reinterpret_cast<const f4*>(acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4);
#pragma unroll
    for (int MT = 0; MT < MT_N; ++MT)
#pragma unroll
      for (int NT = 0; NT < NT_N; ++NT)
        *reinterpret_cast<f4*>(acc + woff + ((MT * NT_N + NT) * 64 + lane) * 4) = cur[MT][NT] + dw[MT][NT];
    if (q == 0) {
#pragma unroll
      for (int MT = 0; MT < MT_N; ++MT) acc[boff + 16 * MT + p] += bs[MT];
    }
  }
}

// Small point sets (T = 1: ONE tile per wave and layer): every tile would send the whole padded gradient through the
// atomic units — 783 tiles x 164 KB at the 12 514 points of train_newmethod.py, 100 of the iteration's 406 us.  The four
// waves of a workgroup are on the same layer (the batch loop is workgroup-uniform), so they add their blocks in LDS first
// and each wave sends ONE of the sum's 16x16 blocks: a quarter of the atomics.  Two buffers alternate between layers, so
// one barrier per flush is enough (a buffer is written again two flushes later, behind the barrier in between, which a
// wave only reaches after its reads).
template <int MT_N, int NT_N, int NACC, int RKS, int CKS>
__device__ __forceinline__ void bwgrad_flush_wg(float* __restrict__ acc, int woff, int boff, const f4 (&dwa)[NACC][MT_N][NT_N],
                                                float (&bs)[MT_N], float* __restrict__ comb, int wave, int lane) {
  const int p = lane & 15, q = lane >> 4;
  constexpr int NTILE = MT_N * NT_N;
  static_assert(NTILE <= BATCH_WAVES, "one block of the sum per wave");
  float* mine = comb + wave * (NTILE * 256 + 64);
#pragma unroll
  for (int MT = 0; MT < MT_N; ++MT) {
    float t = bs[MT];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (q == 0) mine[NTILE * 256 + 16 * MT + p] = t;
#pragma unroll
    for (int NT = 0; NT < NT_N; ++NT) {
      f4 v = dwa[0][MT][NT];
#pragma unroll
      for (int a = 1; a < NACC; ++a) v += dwa[a][MT][NT];
      *reinterpret_cast<f4*>(mine + ((MT * NT_N + NT) * 64 + lane) * 4) = v;
    }
  }
  __syncthreads();
  if (wave < NTILE) {
    const int MT = wave / NT_N, NT = wave % NT_N;
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < BATCH_WAVES; ++w) v += *reinterpret_cast<const f4*>(comb + w * (NTILE * 256 + 64) + (wave * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (4 * MT + r < RKS && (4 * NT + 4 <= CKS || 4 * NT + (p & 3) < CKS))
        __hip_atomic_fetch_add(acc + woff + (wave * 4 + r) * 64 + lane, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (wave == BATCH_WAVES - 1 && lane < 16 * MT_N) {      // the bias rows: 16 MT_N sums of four
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < BATCH_WAVES; ++w) t += comb[w * (NTILE * 256 + 64) + NTILE * 256 + lane];
    __hip_atomic_fetch_add(acc + boff + lane, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__host__ __device__ constexpr int batch_comb_floats(int WP) { return 2 * BATCH_WAVES * ((WP / 16) * (WP / 16) * 256 + 64); }   // two buffers

// WP: padded hidden width (16 / 32); KS = ceil(W / 4): k-steps of a hidden contraction; KS0 = ceil(d_in / 4);
// T: tiles per wave and batch.  Gradient passes only (the forward-only calls stay on k_fused).
template <int WP, int KS, int KS0, int K1, int T, int SINK, int ACT, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(BATCH_THREADS, batch_occ(WP, K1)) void k_fused_batch(const FusedParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTH = WP / 16;
  static_assert(KS <= 4 * NTH && KS > 4 * (NTH - 1), "KS must land in the last 16-feature tile of WP");
    // the wave index as a SCALAR: every spill-slot / pad base below is then an SGPR base + one per-lane offset + an
  // immediate, instead of a 64-bit VGPR pair per (tile, quantity, chunk) — hoisted out of the layer loop those alone
  // overflowed the register file (300 spilled VGPRs at T = 8)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;
  float* lacc = smem;
  constexpr int NPADS = batch_pads(WP, K1);       // per wave: every quantity's Z and A tiles of one layer
  constexpr int NACC = NTH == 1 ? 2 : 1;          // accumulator sets of the weight gradient (bwg_mfma)
  float* tb = smem + P.lds_acc_floats + wave * (NPADS * TB_FLOATS);
  float* lsum = smem + P.lds_acc_floats + BATCH_WAVES * NPADS * TB_FLOATS;
  constexpr bool WGFLUSH = T == 1 && SINK == BSINK_ATOMIC;      // bwgrad_flush_wg
  float* comb = lsum + BATCH_WAVES * MAX_SUMS;                   // (two buffers of batch_comb_floats / 2 each; WGFLUSH only)
  int flushes = 0;
  const int PP = P.PW + P.PB;
  // this wave's gradient sink (see bwgrad_flush): its own LDS copy, or one of the shared copies in HBM
  float* __restrict__ gacc = SINK == BSINK_LDS_WAVE ? lacc + wave * PP
                                                    : P.wg_grads + (int64_t)(blockIdx.x % BATCH_ATOMIC_COPIES) * PP;
  if (SINK == BSINK_LDS_WAVE) {
    for (int i = threadIdx.x; i < BATCH_WAVES * PP; i += BATCH_THREADS) lacc[i] = 0.f;
    __syncthreads();
  }
  float sums[MAX_SUMS];
#pragma unroll
  for (int j = 0; j < MAX_SUMS; ++j) sums[j] = 0.f;
  ScatterMap<K1> sm, sm_mse;
  build_scatter_maps<K1>(P, q, sm, sm_mse);

  const int gw = blockIdx.x * BATCH_WAVES + wave;
  constexpr int SLOTF = K1 * KS * 64;                       // floats of one (tile, layer) spill slot
  const int L = P.L;
  float* __restrict__ scr = P.scratch + (int64_t)gw * P.scratch_per_wave;    // [t][l - 1][SLOTF]
  const int64_t tstride = (int64_t)(L > 1 ? L - 1 : 1) * SLOTF;
  const float* __restrict__ Wp_ = P.Wp;
  const float* __restrict__ WTp_ = P.WTp;
  const float* __restrict__ Bp_ = P.Bp;
  const int64_t n_batches = (P.n_tiles + T - 1) / T;

  // Workgroup-uniform trip count (the waves of a workgroup meet at barriers when WGFLUSH): a wave whose batch lies past
  // the end works on clamped points and contributes exact zeros, like the tiles past the end of a ragged batch.
  for (int64_t bg = blockIdx.x; bg * BATCH_WAVES < n_batches; bg += gridDim.x) {
    const int64_t tile0 = (bg * BATCH_WAVES + wave) * T;
    float a[T][K1][KS];          // the T jets: a_l going up, zbar_l coming down
    // ---- input layer: x at k-step s, lane group q  <->  column 4s + q ------------------------------------------
    float xin[T][KS0];
    static_for<0, T>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      int64_t pc = (tile0 + t) * 16 + p;
      pc = pc < P.N ? pc : P.N - 1;
#pragma unroll
      for (int s = 0; s < KS0; ++s) xin[t][s] = (4 * s + q < P.d_in) ? P.X[pc * P.d_in + 4 * s + q] : 0.f;
    });
    float tang[K1][KS0];         // unit tangents (the same for every tile)
#pragma unroll
    for (int c = 1; c < K1; ++c)
#pragma unroll
      for (int s = 0; s < KS0; ++s) tang[c][s] = (4 * s + q == P.dir_col[c - 1]) ? 1.f : 0.f;
    {
      f4 w0[NTH][1];
      bload_w<1, NTH>(Wp_, 16, w0, p, q);
      f4 bias[NTH];
      load_bias<NTH>(Bp_ + b_off_p<WP>(0), bias, q);
      // GEMM of tile t+1, then the activation of tile t (whose accumulators are long done): order pinned tile by tile —
      // left alone the scheduler runs all T GEMMs first and keeps T accumulator sets live (the register file overflows)
      f4 acc[2][K1][NTH];
      auto gemm0 = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        float b0[K1][KS0];
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
          b0[0][s] = xin[t][s];
#pragma unroll
          for (int c = 1; c < K1; ++c) b0[c][s] = tang[c][s];
        }
        zero_tiles<NTH, K1>(acc[t & 1]);
        bgemm<KS0, 1, NTH, K1>(w0, b0, acc[t & 1]);
      };
      gemm0(std::integral_constant<int, 0>());
      static_for<0, T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 1 < T) gemm0(std::integral_constant<int, t + 1>());
        __builtin_amdgcn_sched_barrier(0);
        bactivate<ACT, KS, NTH, K1>(acc[t & 1], bias, a[t]);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- hidden layers, layer-major over the batch -----------------------------------------------------------
    f4 w[NTH][NTH];
    if (L > 1) bload_w<NTH, NTH>(Wp_ + w_off_p<WP>(1), WP, w, p, q);
    for (int l = 1; l < L; ++l) {
      f4 wn[NTH][NTH];           // next layer's weights, requested a whole batch-layer ahead (last trip: this layer's again)
      bload_w<NTH, NTH>(Wp_ + w_off_p<WP>(l + 1 < L ? l + 1 : l), WP, wn, p, q);
      f4 bias[NTH];
      load_bias<NTH>(Bp_ + b_off_p<WP>(l), bias, q);
      f4 acc[2][K1][NTH];
      auto gemm = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (!(PINN_BATCH_SKIP & 1))
        bspill<KS, K1>(scr + t * tstride + (int64_t)(l - 1) * SLOTF, a[t], lane);    // a_l, re-read by the reverse sweep
        zero_tiles<NTH, K1>(acc[t & 1]);
        bgemm<KS, NTH, NTH, K1>(w, a[t], acc[t & 1]);
      };
      gemm(std::integral_constant<int, 0>());
      static_for<0, T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 1 < T) gemm(std::integral_constant<int, t + 1>());
        __builtin_amdgcn_sched_barrier(0);
        bactivate<ACT, KS, NTH, K1>(acc[t & 1], bias, a[t]);
      });
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int MT = 0; MT < NTH; ++MT)
#pragma unroll
        for (int kt = 0; kt < NTH; ++kt) w[MT][kt] = wn[MT][kt];
    }
    // ---- output layer + loss + its adjoint, tile by tile: a[t] <- zbar_L(t) --------------------------------------
    {
      f4 wo[1][NTH], wtl[NTH][1];
      bload_w<NTH, 1>(Wp_ + w_off_p<WP>(L), WP, wo, p, q);
      bload_w<1, NTH>(WTp_ + w_off_p<WP>(L), 16, wtl, p, q);
      f4 bias_o[1];
      load_bias<1>(Bp_ + b_off_p<WP>(L), bias_o, q);
      f4 dwl[NACC][1][NTH];
      float bsl[1];
      bwg_zero<1, NTH, NACC>(dwl, bsl);
      static_for<0, T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        __builtin_amdgcn_sched_barrier(0);
        const int64_t pt = (tile0 + t) * 16 + p;
        const bool valid = pt < P.N;
        const int64_t ptc = valid ? pt : P.N - 1;
        f4 out[K1][1];
        zero_tiles<1, K1>(out);
        bgemm<KS, NTH, 1, K1>(wo, a[t], out);
        out[0][0] += bias_o[0];
        f4 G[K1][1];
        loss_epilogue<K1, true, true, EPI>(P, out, G, sums, sm, sm_mse, tb, pt, ptc, valid, p, q);
#pragma unroll
        for (int c = 0; c < K1; ++c) {
          btr_write(tb + c * (1 + NTH) * TB_FLOATS, G[c][0], p, q);
          bwg_write_ks<NTH, KS>(tb + (c * (1 + NTH) + 1) * TB_FLOATS, a[t][c], p, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 g[K1][NTH];
        zero_tiles<NTH, K1>(g);
        {   // abar_L = W_L^T G: contraction over the (naturally ordered) outputs, all four k-steps
          float gb[K1][4];
#pragma unroll
          for (int c = 0; c < K1; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) gb[c][r] = G[c][0][r];
          bgemm<4, 1, NTH, K1>(wtl, gb, g);
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 gtr[2][1], atr[2][NTH];
        bwg_read_q<1, NTH>(gtr[0], atr[0], tb, p, q);
        __builtin_amdgcn_sched_barrier(0);
        badjoint<ACT, KS, NTH, K1>(g, a[t], a[t]);
        bwg_tail<1, NTH, K1, NACC>(dwl, bsl, gtr, atr, tb, p, q);
      });
      if constexpr (WGFLUSH) bwgrad_flush_wg<1, NTH, NACC, 4, KS>(gacc, w_off_p<WP>(L), P.PW + b_off_p<WP>(L), dwl, bsl, comb + (flushes++ & 1) * (batch_comb_floats(WP) / 2), wave, lane);
      else bwgrad_flush<1, NTH, NACC, SINK, 4, KS>(gacc, w_off_p<WP>(L), P.PW + b_off_p<WP>(L), dwl, bsl, lane);
    }
    // ---- reverse sweep, layer-major: abar_l = W_l^T zbar_l ; dW_l += zbar_l (x) a_l ; zbar_{l-1} = adjoint ---------
    if (L > 1) bload_w<NTH, NTH>(WTp_ + w_off_p<WP>(L - 1), WP, w, p, q);
    // a_l of the next PF tiles is always in flight (a ring of PF register sets): one tile ahead left ~15 KB per CU in
    // flight, and with it the reverse sweep waiting on HBM latency (co100x20: 24.7 ms with the loads, 15.6 without)
    constexpr int PF = BATCH_PF < T ? BATCH_PF : T;
    static_assert(T % PF == 0, "the ring position of tile t must not depend on the layer");
    // Every load of this loop is UNCONDITIONAL (the last layer re-reads its own slots / weights instead of skipping):
    // a load behind a runtime branch makes the compiler's vmcnt bookkeeping assume it was not issued, and the waits
    // in front of each tile's a_l collapsed to vmcnt(0) — the ring drained every tile.
    float pf[PF][K1][KS];
    static_for<0, PF>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      bunspill<KS, K1>(scr + t * tstride + (int64_t)(L > 1 ? L - 2 : 0) * SLOTF, pf[t], lane);
    });
    for (int l = L - 1; l >= 1; --l) {
      const int ln = l >= 2 ? l - 1 : 1;        // the next layer down (itself on the last trip: loaded, never used)
      f4 wn[NTH][NTH];
      bload_w<NTH, NTH>(WTp_ + w_off_p<WP>(ln), WP, wn, p, q);
      f4 dw[NACC][NTH][NTH];
      float bs[NTH];
      bwg_zero<NTH, NTH, NACC>(dw, bs);
      static_for<0, T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        __builtin_amdgcn_sched_barrier(0);
        float (&ai)[K1][KS] = pf[t % PF];      // a_l of this tile
        // the phase order below IS the latency plan (pad writes | GEMM | pad reads | VALU | MFMAs): pinned, because the
        // machine scheduler otherwise sinks every ds_read next to the MFMA that consumes it (an lgkmcnt wait in front
        // of each of the 48 MFMAs)
#pragma unroll
        for (int c = 0; c < K1; ++c) {     // operands of dW_l on their way through the pads while the matrix pipe runs abar_l
          bwg_write_ks<NTH, KS>(tb + c * 2 * NTH * TB_FLOATS, a[t][c], p, q);
          bwg_write_ks<NTH, KS>(tb + (c * 2 * NTH + NTH) * TB_FLOATS, ai[c], p, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 g2[K1][NTH];
        zero_tiles<NTH, K1>(g2);
        bgemm<KS, NTH, NTH, K1>(w, a[t], g2);
        __builtin_amdgcn_sched_barrier(0);
        f4 ztr[2][NTH], atr[2][NTH];
        bwg_read_q<NTH, NTH>(ztr[0], atr[0], tb, p, q);
        __builtin_amdgcn_sched_barrier(0);
        badjoint<ACT, KS, NTH, K1>(g2, ai, a[t]);      // (VALU: covers the transposed reads)
        __builtin_amdgcn_sched_barrier(0);
        // ai is dead: its registers take tile t + PF of this layer, or tile t + PF - T of the next one down
        if constexpr (!(PINN_BATCH_SKIP & 2)) {
          if constexpr (t + PF < T) bunspill<KS, K1>(scr + (t + PF) * tstride + (int64_t)(l - 1) * SLOTF, pf[t % PF], lane);
          else bunspill<KS, K1>(scr + (t + PF - T) * tstride + (int64_t)(ln - 1) * SLOTF, pf[t % PF], lane);
        }
        bwg_tail<NTH, NTH, K1, NACC>(dw, bs, ztr, atr, tb, p, q);
      });
      if constexpr (!(PINN_BATCH_SKIP & 32))
      {
      if constexpr (WGFLUSH) bwgrad_flush_wg<NTH, NTH, NACC, KS, KS>(gacc, w_off_p<WP>(l), P.PW + b_off_p<WP>(l), dw, bs, comb + (flushes++ & 1) * (batch_comb_floats(WP) / 2), wave, lane);
      else bwgrad_flush<NTH, NTH, NACC, SINK, KS, KS>(gacc, w_off_p<WP>(l), P.PW + b_off_p<WP>(l), dw, bs, lane);
      }
      else {   // diagnostic: no flush, but every accumulator stays live
        float acc_ = bs[0];
#pragma unroll
        for (int a_ = 0; a_ < NACC; ++a_)
#pragma unroll
          for (int MT = 0; MT < NTH; ++MT)
#pragma unroll
            for (int NT = 0; NT < NTH; ++NT) acc_ += (dw[a_][MT][NT][0] + dw[a_][MT][NT][1]) + (dw[a_][MT][NT][2] + dw[a_][MT][NT][3]);
        sums[MAX_SUMS - 1] += acc_;
      }
#pragma unroll
      for (int MT = 0; MT < NTH; ++MT)
#pragma unroll
        for (int kt = 0; kt < NTH; ++kt) w[MT][kt] = wn[MT][kt];
    }
    // ---- layer 0: dW_0 += zbar_0 (x) (x, unit tangents) ---------------------------------------------------------
    {
      f4 dw0[NACC][NTH][1];
      float bs0[NTH];
      bwg_zero<NTH, 1, NACC>(dw0, bs0);
      static_for<0, T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < K1; ++c) {
          bwg_write_ks<NTH, KS>(tb + c * (NTH + 1) * TB_FLOATS, a[t][c], p, q);
          f4 xt;
#pragma unroll
          for (int r = 0; r < 4; ++r) xt[r] = r < KS0 ? (c == 0 ? xin[t][r < KS0 ? r : 0] : tang[c][r < KS0 ? r : 0]) : 0.f;
          btr_write(tb + (c * (NTH + 1) + NTH) * TB_FLOATS, xt, p, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 ztr[2][NTH], xtr[2][1];
        bwg_read_q<NTH, 1>(ztr[0], xtr[0], tb, p, q);
        bwg_tail<NTH, 1, K1, NACC>(dw0, bs0, ztr, xtr, tb, p, q);
      });
      if constexpr (WGFLUSH) bwgrad_flush_wg<NTH, 1, NACC, KS, KS0>(gacc, 0, P.PW + b_off_p<WP>(0), dw0, bs0, comb + (flushes++ & 1) * (batch_comb_floats(WP) / 2), wave, lane);
      else bwgrad_flush<NTH, 1, NACC, SINK, KS, KS0>(gacc, 0, P.PW + b_off_p<WP>(0), dw0, bs0, lane);
    }
  }

  // ---- per-workgroup reductions (as k_fused) --------------------------------------------------------------------
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
    for (int w2 = 0; w2 < BATCH_WAVES; ++w2) v += lsum[w2 * MAX_SUMS + threadIdx.x];
    P.wg_sums[(int64_t)blockIdx.x * MAX_SUMS + threadIdx.x] = v;
  }
  if (SINK == BSINK_LDS_WAVE) {   // (the __syncthreads above covers the waves' last flushes)
    float* dst = P.wg_grads + (int64_t)blockIdx.x * PP;
    for (int i = threadIdx.x; i < PP; i += BATCH_THREADS)
      dst[i] = (lacc[i] + lacc[PP + i]) + (lacc[2 * PP + i] + lacc[3 * PP + i]);
  }
}

// launchers: one translation unit per padded width (pinn_fused_batch_w16.hip / _w32.hip)
template <int WP>
int launch_fused_batch(int W, int d_in, int K1, const FusedParams& P, int grid, size_t lds_bytes, hipStream_t s);
bool fused_batch_has_kernel(int WP, int W, int d_in, int K1, int act);

}  // namespace pinn
