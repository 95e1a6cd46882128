// wide_kernel.h — MFMA engine for hidden widths 64 < W <= 256 (BASELINE configs[3]: 12 x 256).
//
// Same algorithm and the same "acc layout" as fused_kernel.h (forward-mode jet chain on
// v_mfma_f32_16x16x4_f32, accumulators feed the next MFMA as B operand unchanged), but a
// 256-wide layer does not fit one wave's registers twice over, so the network is walked ONE
// LAYER PER LAUNCH:
//   k_wide_fwd   per 16-point tile: out[K1][NTO] = act(W . in + b); the input jet streams through
//                registers in 64-feature chunks, the NTO*K1 accumulator tiles (256 VGPR for
//                16 x 4) stay resident; weights come straight from L2 (every wave of the launch
//                works on the same layer, so the 256 KB matrix is L2-hot).
//   k_wide_bwd   zbar = activation adjoint (in place), abar_in = W^T zbar (same chain on W^T).
//   k_wide_wgrad split-K weight gradient: a wave owns a (64 x in) block of dW in registers and
//                walks a strided subset of the point tiles (transposing each 16x16 block through
//                its LDS pads exactly as the fused kernel does); one global atomic flush per wave.
// Activations live in HBM in fragment-native order [tile][quantity][feature tile][lane][4]
// (16 B per lane, fully coalesced both ways).  The host loops over point CHUNKS so the
// activation workspace stays bounded (pinn_wide.hip).
#pragma once
#include "fused_kernel.h"

namespace pinn {

// weight-gradient operand pads: fp32 mode uses fused_kernel.h's feature-major layout (one conflict-free ds_read_b128 per
// block, PINN_FUSED_TR2); bf16 mode keeps the first layout (its operands are packed pairwise from consecutive points)
#ifndef PINN_WIDE_TR2
#define PINN_WIDE_TR2 PINN_FUSED_TR2
#endif
template <bool BF16>
__device__ __forceinline__ void wtr_write(float* __restrict__ tb, f4 v, int p, int q) {
  if constexpr (!BF16 && PINN_WIDE_TR2) transpose_write2(tb, v, p, q);
  else transpose_write(tb, v, p, q);
}
template <bool BF16>
__device__ __forceinline__ f4 wtr_read(const float* __restrict__ tb, int p, int q) {
  if constexpr (!BF16 && PINN_WIDE_TR2) return transpose_read2(tb, p, q);
  else return transpose_read<BF16>(tb, p, q);
}


constexpr int WIDE_WAVES = 4;
constexpr int WIDE_THREADS = WIDE_WAVES * 64;
constexpr int WIDE_MAX_PADS = 20;   // 4 zbar tiles + 16 input tiles of one quantity
#ifndef PINN_WIDE_SHARE_A
#define PINN_WIDE_SHARE_A 1
#endif

struct WideLayer {
  const float* W;        // this layer's padded weights, row-major [16*NTO][16*NTI] (fwd) or W^T (bwd)
  const unsigned short* W16;  // the same matrix in bf16 (BF16 kernels)
  const float* b;        // padded bias (fwd)
  const float* in_act;   // fwd: layer input jet; bwd: a_{l+1} (layer output jet); wgrad: layer input jet
  float* out_act;        // fwd: layer output jet
  float* g_in;           // bwd: adjoint of the layer output (overwritten with zbar); wgrad: zbar
  float* g_out;          // bwd: adjoint of the layer input
  float* z_out;          // bwd: zbar (activation adjoint applied), read by wgrad as g_in
  float* dW;             // wgrad: flat torch-layout gradient of this layer's weight (out_d x in_d)
  float* db;             // wgrad: gradient of this layer's bias
  int in_d, out_d;       // real (unpadded) dims of the layer
  int64_t tile0;         // first global tile of this chunk (for X / T indexing)
  int64_t n_tiles;       // tiles in this chunk
  int sums_slot;         // row offset in wg_sums for this chunk
};

// ---- bf16 operands (PINN_PREC_BF16): v_mfma_f32_16x16x16_bf16 takes k = 4*(lane>>4) + j, j = 0..3,
// per lane — exactly the 4 registers of one accumulator tile (B operand) and 8 contiguous bytes of
// a bf16 weight row (A operand).  Accumulation stays fp32.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 to_bf16x4(f4 v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};   // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
  return __builtin_bit_cast(bf16x4, b);
}
__device__ __forceinline__ f4 mfma_bf16(bf16x4 a, bf16x4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 load_w16(const unsigned short* __restrict__ W16, int idx) {
  return *reinterpret_cast<const bf16x4*>(W16 + idx);
}

// ---- jet storage format.  In bf16 mode the per-layer kernels are bound by the HBM traffic of the jets
// (512 KB per 16-point tile and layer in fp32), and every stored jet is only ever consumed as a bf16
// MFMA operand or by the activation adjoint, so they are STORED as bf16 too (half the traffic).  The
// output adjoint G of the (fp32) output layer stays fp32.  FMT bits per kernel:
constexpr int FMT_IN16 = 1;    // Lp.in_act holds bf16
constexpr int FMT_GIN16 = 2;   // Lp.g_in holds bf16
constexpr int FMT_OUT16 = 4;   // what the kernel writes (out_act / g_out / z_out) is bf16
// "chain layout" (chain_kernel.h): bf16 jets of the bf16-mode engine, [tile][quantity][k-step s][point][32 units];
// the first / last layer's kernels of this file read and write the chain kernels' buffers in it
constexpr int FMT_IN_NEW = 8, FMT_GIN_NEW = 16, FMT_OUT_NEW = 32;
__device__ __forceinline__ f4 from_bf16x4(bf16x4 v) {
  f4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __builtin_bit_cast(float, (unsigned)(unsigned short)v[i] << 16);
  return o;
}
// block = one 16x16 tile of one quantity (256 elements), fragment-native: element lane*4 + r.
// blk = (tile * K1 + quantity) * NT + MT.  NEWL (bf16 only, NT even): the chain layout — tile MT of a quantity is
// the h = MT & 1 half of k-step s = MT >> 1: 4 elements at ((tq * NT/2 + s) * 512 + (4p + q) * 8 + 4h.
template <int NT>
__device__ __forceinline__ int64_t jet_off_new(int64_t blk, int lane) {
  const int64_t tq = blk / NT;
  const int MT = (int)(blk % NT);
  return (tq * (NT / 2) + (MT >> 1)) * 512 + (4 * (lane & 15) + (lane >> 4)) * 8 + (MT & 1) * 4;
}
template <bool H, bool NEWL = false, int NT = 2>
__device__ __forceinline__ f4 jet_ld(const float* base, int64_t blk, int lane) {
  if constexpr (NEWL) {
    static_assert(!NEWL || (H && NT % 2 == 0), "chain layout: bf16 jets of whole k-steps");
    return from_bf16x4(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const unsigned short*>(base) + jet_off_new<NT>(blk, lane)));
  } else if constexpr (H) return from_bf16x4(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const unsigned short*>(base) + blk * 256 + lane * 4));
  else return *reinterpret_cast<const f4*>(base + blk * 256 + lane * 4);
}
template <bool H, bool NEWL = false, int NT = 2>
__device__ __forceinline__ void jet_st(float* base, int64_t blk, int lane, f4 v) {
  if constexpr (NEWL) {
    static_assert(!NEWL || (H && NT % 2 == 0), "chain layout: bf16 jets of whole k-steps");
    *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned short*>(base) + jet_off_new<NT>(blk, lane)) = to_bf16x4(v);
  } else if constexpr (H) *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned short*>(base) + blk * 256 + lane * 4) = to_bf16x4(v);
  else *reinterpret_cast<f4*>(base + blk * 256 + lane * 4) = v;
}

// acc[c][MT] += W[16MT + m][16(kc+j) ..] . B[c][j] for every output tile MT of one input chunk.
// The weight fragments of tile MT+1 are requested before the MFMAs of tile MT issue (one wave per
// SIMD: an L2 round trip per output tile would otherwise sit exposed 16 times per chunk).
// `mid`: the caller's zbar stores (k_wide_bwd), issued here rather than by the caller so that in bf16 mode
// they go out after the operand conversions (4 % faster; fp32 keeps them in front of the chunk).
template <int CH, int K1, int NTO, bool BF16, class Mid = NoMid>
__device__ __forceinline__ void wide_mac_chunk(const WideLayer& Lp, int kc, int ldw, int p, int q, const f4 (&B)[K1][CH],
                                               f4 (&acc)[K1][NTO], const Mid& mid = Mid()) {
  if constexpr (BF16) {
    bf16x4 B16[K1][CH];
#pragma unroll
    for (int c = 0; c < K1; ++c)
#pragma unroll
      for (int j = 0; j < CH; ++j) B16[c][j] = to_bf16x4(B[c][j]);
    mid();
    // (no look-ahead here: measured 12 % slower in bf16 mode, where the MFMA block per tile is short)
#pragma unroll
    for (int MT = 0; MT < NTO; ++MT) {
      bf16x4 a[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] = load_w16(Lp.W16, (16 * MT + p) * ldw + 16 * (kc + j) + 4 * q);
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int c = 0; c < K1; ++c) acc[c][MT] = mfma_bf16(a[j], B16[c][j], acc[c][MT]);
    }
  } else {
    mid();   // (fp32: in front of the chunk — from inside it, behind output tile 1's weights, measured 2 % slower)
    f4 a[2][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) a[0][j] = *reinterpret_cast<const f4*>(Lp.W + p * ldw + 16 * (kc + j) + 4 * q);
#pragma unroll
    for (int MT = 0; MT < NTO; ++MT) {
      if (MT + 1 < NTO) {
#pragma unroll
        for (int j = 0; j < CH; ++j)
          a[(MT + 1) & 1][j] = *reinterpret_cast<const f4*>(Lp.W + (16 * (MT + 1) + p) * ldw + 16 * (kc + j) + 4 * q);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < K1; ++c) acc[c][MT] = mfma4(a[MT & 1][j][r], B[c][j][r], acc[c][MT]);
    }
  }
}

template <int K1>
__device__ __forceinline__ void wide_input_jet(const FusedParams& P, int64_t ptc, int q, f4 (&b)[K1][1]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 4 * q + r;
    b[0][0][r] = (f < P.d_in) ? P.X[ptc * P.d_in + f] : 0.f;
#pragma unroll
    for (int c = 1; c < K1; ++c) b[c][0][r] = (f == P.dir_col[c - 1]) ? 1.f : 0.f;
  }
}

// out = act(W . in + b) for one layer.  FIRST: the input is (x, unit tangents) built from X.
// LAST: no activation; outputs / loss / output adjoint instead of a stored jet.
// HV = 2: the output tiles of one 16-point tile are split between two waves (NTO/2 accumulator tiles
// = 128 registers each), so two waves fit per SIMD and cover each other's load latencies; both read
// the same input jet (second read served by L1/L2).
template <int NTI, int NTO_ALL, int K1, int ACT, bool FIRST, bool LAST, bool GRAD, bool BF16, int HV = 1, int FMT = 0>
__global__ __launch_bounds__(WIDE_THREADS * HV, HV) void k_wide_fwd(const FusedParams P, const WideLayer Lq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) / HV, hf = (threadIdx.x >> 6) % HV;
  const int p = lane & 15, q = lane >> 4;
  constexpr int NTO = NTO_ALL / HV;
  WideLayer Lp = Lq;
  Lp.W = Lq.W + hf * NTO * 16 * (16 * NTI);
  Lp.W16 = Lq.W16 + hf * NTO * 16 * (16 * NTI);
  Lp.b = Lq.b + hf * NTO * 16;
  float* tb = smem + wave * (WIDE_MAX_PADS * TB_FLOATS);
  float* lsum = smem + WIDE_WAVES * WIDE_MAX_PADS * TB_FLOATS;
  constexpr int CH = NTI < 4 ? NTI : 4;
  constexpr int LDW = 16 * NTI;
  float sums[MAX_SUMS];
#pragma unroll
  for (int j = 0; j < MAX_SUMS; ++j) sums[j] = 0.f;
  ScatterMap<K1> sm, sm_mse;
  if constexpr (LAST) build_scatter_maps<K1>(P, q, sm, sm_mse);
  const int gw = blockIdx.x * WIDE_WAVES + wave, nw = gridDim.x * WIDE_WAVES;
  for (int64_t t = gw; t < Lp.n_tiles; t += nw) {
    const int64_t pt = (Lp.tile0 + t) * 16 + p;
    const bool valid = pt < P.N;
    const int64_t ptc = valid ? pt : P.N - 1;
    f4 acc[K1][NTO];
    init_bias<NTO, K1>(Lp.b, acc, q);
    auto load_chunk = [&](int kc, f4 (&Bc)[K1][CH]) {
#pragma unroll
      for (int c = 0; c < K1; ++c)
#pragma unroll
        for (int j = 0; j < CH; ++j)
          Bc[c][j] = jet_ld<(FMT & FMT_IN16) != 0, (FMT & FMT_IN_NEW) != 0, (NTI > 1 ? NTI : 2)>(Lp.in_act, (t * K1 + c) * NTI + kc + j, lane);
    };
    for (int kc = 0; kc < NTI; kc += CH) {
      f4 B[K1][CH];
      if constexpr (FIRST) {
        f4 b0[K1][1];
        wide_input_jet<K1>(P, ptc, q, b0);
#pragma unroll
        for (int c = 0; c < K1; ++c) B[c][0] = b0[c][0];
      } else {
        load_chunk(kc, B);
      }
      wide_mac_chunk<CH, K1, NTO, BF16>(Lp, kc, LDW, p, q, B, acc);
    }
    if constexpr (!LAST) {
      activate<ACT, NTO, K1>(acc);
#pragma unroll
      for (int c = 0; c < K1; ++c)
#pragma unroll
        for (int MT = 0; MT < NTO; ++MT)
          jet_st<(FMT & FMT_OUT16) != 0, (FMT & FMT_OUT_NEW) != 0, (NTO_ALL > 1 ? NTO_ALL : 2)>(Lp.out_act, (t * K1 + c) * NTO_ALL + hf * NTO + MT, lane, acc[c][MT]);
    } else {
      static_assert(!LAST || NTO == 1, "the output layer has one (padded) tile");
      f4 (&out)[K1][1] = acc;
      f4 G[K1][1];
      loss_epilogue<K1, GRAD>(P, out, G, sums, sm, sm_mse, tb, pt, ptc, valid, p, q);
      if constexpr (GRAD) {
#pragma unroll
        for (int c = 0; c < K1; ++c) *reinterpret_cast<f4*>(Lp.g_out + ((t * K1 + c) * 1 + 0) * 256 + lane * 4) = G[c][0];
      }
    }
  }
  if constexpr (LAST) {
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
      for (int w = 0; w < WIDE_WAVES; ++w) v += lsum[w * MAX_SUMS + threadIdx.x];
      P.wg_sums[((int64_t)Lp.sums_slot + blockIdx.x) * MAX_SUMS + threadIdx.x] = v;
    }
  }
}

// zbar = adjoint through the activation (HIDDEN; written back over g_in), abar_in = W^T zbar (NEED_GIN).
// NTK = tiles of the layer OUTPUT (the contraction axis here), NTO = tiles of the layer INPUT.
// HV = 2 as in k_wide_fwd (input-feature tiles split between two waves).  zbar goes to z_out, never
// back over g_in: with two waves per tile an in-place update would be read twice.
template <int NTK, int NTO_ALL, int K1, int ACT, bool HIDDEN, bool NEED_GIN, bool BF16, int HV = 1, int FMT = 0>
__global__ __launch_bounds__(WIDE_THREADS * HV, HV) void k_wide_bwd(const FusedParams P, const WideLayer Lq) {
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) / HV, hf = (threadIdx.x >> 6) % HV;
  const int p = lane & 15, q = lane >> 4;
  constexpr int NTO = NTO_ALL / HV;
  WideLayer Lp = Lq;
  if (NEED_GIN) {
    Lp.W = Lq.W + hf * NTO * 16 * (16 * NTK);
    Lp.W16 = Lq.W16 + hf * NTO * 16 * (16 * NTK);
  }
  constexpr int CH = NTK < 4 ? NTK : 4;
  constexpr int LDW = 16 * NTK;
  const int gw = blockIdx.x * WIDE_WAVES + wave, nw = gridDim.x * WIDE_WAVES;
  // local restrict-qualified views: the jets, the incoming adjoint and the two outputs never overlap
  const float* __restrict__ gin_p = Lp.g_in;
  const float* __restrict__ act_p = Lp.in_act;
  float* __restrict__ zout_p = (HV == 1) ? Lp.g_in : Lp.z_out;
  float* __restrict__ gout_p = Lp.g_out;
  for (int64_t t = gw; t < Lp.n_tiles; t += nw) {
    f4 acc[K1][NTO];
    zero_tiles<NTO, K1>(acc);
    for (int kc = 0; kc < NTK; kc += CH) {
      f4 g[K1][CH];
#pragma unroll
      for (int c = 0; c < K1; ++c)
#pragma unroll
        for (int j = 0; j < CH; ++j)
          g[c][j] = jet_ld<(FMT & FMT_GIN16) != 0, (FMT & FMT_GIN_NEW) != 0, (NTK > 1 ? NTK : 2)>(HV == 1 ? Lp.g_in : gin_p, (t * K1 + c) * NTK + kc + j, lane);
      if constexpr (HIDDEN) {
        f4 ao[K1][CH];
#pragma unroll
        for (int c = 0; c < K1; ++c)
#pragma unroll
          for (int j = 0; j < CH; ++j)
            ao[c][j] = jet_ld<(FMT & FMT_IN16) != 0, (FMT & FMT_IN_NEW) != 0, (NTK > 1 ? NTK : 2)>(act_p, (t * K1 + c) * NTK + kc + j, lane);
        activate_adjoint<ACT, CH, K1>(g, ao);
        // HV == 1: in place through the SAME pointer the loads use (the compiler then knows the store
        // cannot alias the next chunk's loads); HV == 2: separate buffer, written by one of the two waves
      }
      // zbar store: handed to the GEMM chunk, which decides where it goes (wide_mac_chunk)
      auto zstore = [&]() {
        if constexpr (HIDDEN) {
          float* zdst = (HV == 1) ? Lp.g_in : zout_p;
          if (hf == 0) {
#pragma unroll
            for (int c = 0; c < K1; ++c)
#pragma unroll
              for (int j = 0; j < CH; ++j)
                jet_st<(FMT & FMT_OUT16) != 0, (FMT & FMT_OUT_NEW) != 0 && (NTK > 1), (NTK > 1 ? NTK : 2)>(zdst, (t * K1 + c) * NTK + kc + j, lane, g[c][j]);
          }
        }
      };
      if constexpr (NEED_GIN) wide_mac_chunk<CH, K1, NTO, BF16>(Lp, kc, LDW, p, q, g, acc, zstore);
      else zstore();
    }
    if constexpr (NEED_GIN) {
#pragma unroll
      for (int c = 0; c < K1; ++c)
#pragma unroll
        for (int MT = 0; MT < NTO; ++MT)
          jet_st<(FMT & FMT_OUT16) != 0, (FMT & FMT_OUT_NEW) != 0, (NTO_ALL > 1 ? NTO_ALL : 2)>(gout_p, (t * K1 + c) * NTO_ALL + hf * NTO + MT, lane, acc[c][MT]);
    }
  }
}

// Split-K weight gradient.  The NTM/MTB row blocks of dW are the WAVES of one workgroup walking
// the SAME point tiles, so the input-jet tiles every row block needs are fetched from HBM once
// and re-read from L1/L2 (as separate workgroups they landed on different XCDs and the kernel ran
// at the HBM roof re-reading them).
// NTM = output tiles of the layer, NTN = input tiles.  FIRST: the layer input is (x, tangents).
template <int MTB, int NTM, int NTN, int K1, bool FIRST, bool BF16, int FMT = 0>
__global__ __launch_bounds__(WIDE_THREADS, 1) void k_wide_wgrad(const FusedParams P, const WideLayer Lp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  float* tb = smem + wave * (WIDE_MAX_PADS * TB_FLOATS);
  constexpr int RB = NTM / MTB;                       // row blocks: 4 (W = 256), 2 (W = 128) or 1 (output layer)
  constexpr int GPW = WIDE_WAVES / RB;                // tile groups per workgroup
  const int rb = wave % RB;
  f4 dw[MTB][NTN];
  float bs[MTB];
#pragma unroll
  for (int MT = 0; MT < MTB; ++MT) {
    bs[MT] = 0.f;
#pragma unroll
    for (int NT = 0; NT < NTN; ++NT) dw[MT][NT] = f4{0.f, 0.f, 0.f, 0.f};
  }
  const int gw = blockIdx.x * GPW + wave / RB, nw = gridDim.x * GPW;
#if PINN_WIDE_SHARE_A
  // W = 256 hidden layers (four row blocks = the four waves, all on the same tile): the 16 input tiles
  // every wave needs are loaded and transposed ONCE per workgroup — each wave handles four of them
  // into a shared, double-buffered set of pads, one barrier per (tile, quantity) — instead of once
  // per wave.  That frees the registers the fp32 kernel lacked for fetching one step ahead (its
  // in-place loads left the HBM latency exposed before every quantity: 40 % of the MFMA rate).
  if constexpr (RB == WIDE_WAVES && !FIRST && NTN % WIDE_WAVES == 0) {
    constexpr int APW = NTN / WIDE_WAVES;
    float* zpad = smem + wave * (4 * TB_FLOATS);
    float* apad = smem + WIDE_WAVES * 4 * TB_FLOATS;          // 2 x NTN pads
    f4 rz[MTB], ra[APW];
    auto fetch = [&](int64_t t, int c) {
#pragma unroll
      for (int MT = 0; MT < MTB; ++MT)
        rz[MT] = jet_ld<(FMT & FMT_GIN16) != 0, (FMT & FMT_GIN_NEW) != 0, (NTM > 1 ? NTM : 2)>(Lp.g_in, (t * K1 + c) * NTM + rb * MTB + MT, lane);
#pragma unroll
      for (int i = 0; i < APW; ++i)
        ra[i] = jet_ld<(FMT & FMT_IN16) != 0, (FMT & FMT_IN_NEW) != 0, (NTN > 1 ? NTN : 2)>(Lp.in_act, (t * K1 + c) * NTN + wave * APW + i, lane);
    };
    if (gw < Lp.n_tiles) fetch(gw, 0);
    int buf = 0;
    for (int64_t t = gw; t < Lp.n_tiles; t += nw) {      // gw, nw are workgroup-uniform here (GPW == 1)
#pragma unroll
      for (int c = 0; c < K1; ++c) {
        float* ap = apad + buf * (NTN * TB_FLOATS);
#pragma unroll
        for (int MT = 0; MT < MTB; ++MT) wtr_write<BF16>(zpad + MT * TB_FLOATS, rz[MT], p, q);
#pragma unroll
        for (int i = 0; i < APW; ++i) wtr_write<BF16>(ap + (wave * APW + i) * TB_FLOATS, ra[i], p, q);
        if (c + 1 < K1) fetch(t, c + 1);
        else if (t + nw < Lp.n_tiles) fetch(t + nw, 0);
        __syncthreads();
        f4 zt[MTB];
#pragma unroll
        for (int MT = 0; MT < MTB; ++MT) zt[MT] = wtr_read<BF16>(zpad + MT * TB_FLOATS, p, q);
        if (c == 0) {
#pragma unroll
          for (int MT = 0; MT < MTB; ++MT) bs[MT] += (zt[MT][0] + zt[MT][1]) + (zt[MT][2] + zt[MT][3]);
        }
#pragma unroll
        for (int N0 = 0; N0 < NTN; N0 += 4) {
          f4 at[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) at[j] = wtr_read<BF16>(ap + (N0 + j) * TB_FLOATS, p, q);
          if constexpr (BF16) {
            bf16x4 a16[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a16[j] = to_bf16x4(at[j]);
#pragma unroll
            for (int MT = 0; MT < MTB; ++MT) {
              const bf16x4 z16 = to_bf16x4(zt[MT]);
#pragma unroll
              for (int j = 0; j < 4; ++j) dw[MT][N0 + j] = mfma_bf16(z16, a16[j], dw[MT][N0 + j]);
            }
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int MT = 0; MT < MTB; ++MT)
#pragma unroll
                for (int j = 0; j < 4; ++j) dw[MT][N0 + j] = mfma4(zt[MT][s], at[j][s], dw[MT][N0 + j]);
          }
        }
        buf ^= 1;
      }
    }
  } else
#endif
  {
  // raw (acc-layout) tiles of ONE quantity, fetched one step ahead of their use: with a single wave
  // per SIMD the HBM/L2 latency of these loads is otherwise fully exposed before every transpose.
  f4 rz[MTB], ra[NTN];
  auto fetch = [&](int64_t t, int c) {
#pragma unroll
    for (int MT = 0; MT < MTB; ++MT)
      rz[MT] = jet_ld<(FMT & FMT_GIN16) != 0, (FMT & FMT_GIN_NEW) != 0, (NTM > 1 ? NTM : 2)>(Lp.g_in, (t * K1 + c) * NTM + rb * MTB + MT, lane);
    if constexpr (FIRST) {
      const int64_t pt = (Lp.tile0 + t) * 16 + p;
      f4 b0[K1][1];
      wide_input_jet<K1>(P, pt < P.N ? pt : P.N - 1, q, b0);
#pragma unroll
      for (int cc = 0; cc < K1; ++cc) ra[0] = (cc == c) ? b0[cc][0] : ra[0];
    } else {
#pragma unroll
      for (int NT = 0; NT < NTN; ++NT)
        ra[NT] = jet_ld<(FMT & FMT_IN16) != 0, (FMT & FMT_IN_NEW) != 0, (NTN > 1 ? NTN : 2)>(Lp.in_act, (t * K1 + c) * NTN + NT, lane);
    }
  };
  // (fp32 mode: the 256 MFMAs per quantity leave no registers for the look-ahead — measured 30 %
  // slower with it — so there the loads are issued in place)
  constexpr bool AHEAD = BF16;
  if (AHEAD && gw < Lp.n_tiles) fetch(gw, 0);
  for (int64_t t = gw; t < Lp.n_tiles; t += nw) {
#pragma unroll
    for (int c = 0; c < K1; ++c) {
      f4 zt[MTB], at[NTN];
      if (!AHEAD) fetch(t, c);
#pragma unroll
      for (int MT = 0; MT < MTB; ++MT) wtr_write<BF16>(tb + MT * TB_FLOATS, rz[MT], p, q);
#pragma unroll
      for (int NT = 0; NT < NTN; ++NT) wtr_write<BF16>(tb + (4 + NT) * TB_FLOATS, ra[NT], p, q);
      // the raw registers are free again: start the next quantity's (or the next tile's) loads now
      if (AHEAD) {
        if (c + 1 < K1) fetch(t, c + 1);
        else if (t + nw < Lp.n_tiles) fetch(t + nw, 0);
      }
      // fp32: element s <-> point 4s + q (one k-step of 16x16x4 per s);
      // bf16: element j <-> point 4q + j (the single k = 16 step of 16x16x16)
#pragma unroll
      for (int MT = 0; MT < MTB; ++MT) zt[MT] = wtr_read<BF16>(tb + MT * TB_FLOATS, p, q);
#pragma unroll
      for (int NT = 0; NT < NTN; ++NT) at[NT] = wtr_read<BF16>(tb + (4 + NT) * TB_FLOATS, p, q);
      if (c == 0) {
#pragma unroll
        for (int MT = 0; MT < MTB; ++MT) bs[MT] += (zt[MT][0] + zt[MT][1]) + (zt[MT][2] + zt[MT][3]);
      }
      if constexpr (BF16) {
        bf16x4 z16[MTB], a16[NTN];
#pragma unroll
        for (int MT = 0; MT < MTB; ++MT) z16[MT] = to_bf16x4(zt[MT]);
#pragma unroll
        for (int NT = 0; NT < NTN; ++NT) a16[NT] = to_bf16x4(at[NT]);
#pragma unroll
        for (int MT = 0; MT < MTB; ++MT)
#pragma unroll
          for (int NT = 0; NT < NTN; ++NT) dw[MT][NT] = mfma_bf16(z16[MT], a16[NT], dw[MT][NT]);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int MT = 0; MT < MTB; ++MT)
#pragma unroll
            for (int NT = 0; NT < NTN; ++NT) dw[MT][NT] = mfma4(zt[MT][s], at[NT][s], dw[MT][NT]);
      }
    }
  }
  }
  // one flush per wave into the flat torch-layout gradient (out_d x in_d row-major)
#pragma unroll
  for (int MT = 0; MT < MTB; ++MT) {
#pragma unroll
    for (int NT = 0; NT < NTN; ++NT)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (rb * MTB + MT) + 4 * q + r, col = 16 * NT + p;
        if (row < Lp.out_d && col < Lp.in_d)
          __hip_atomic_fetch_add(Lp.dW + (int64_t)row * Lp.in_d + col, dw[MT][NT][r], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
    float t = bs[MT];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    const int row = 16 * (rb * MTB + MT) + p;
    if (q == 0 && row < Lp.out_d)
      __hip_atomic_fetch_add(Lp.db + row, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// launchers, one translation unit per padded width (pinn_wide_w128.hip / _w256.hip)
template <int NTW>
int launch_wide_fwd(int which /*0 first, 1 hidden, 2 last*/, int K1, int prec, bool grad, const FusedParams& P,
                    const WideLayer& Lp, int grid, hipStream_t s);
template <int NTW>
int launch_wide_bwd(int which /*0 first, 1 hidden, 2 last*/, int K1, int prec, const FusedParams& P, const WideLayer& Lp,
                    int grid, hipStream_t s);
template <int NTW>
int launch_wide_wgrad(int which /*0 first, 1 hidden, 2 last*/, int K1, int prec, const FusedParams& P, const WideLayer& Lp,
                      int grid_x, hipStream_t s);

}  // namespace pinn
