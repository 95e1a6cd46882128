// chain_kernel.h — bf16-MFMA engine for hidden widths 64 < W <= 256 (BASELINE configs[3]: 12 x 256,
// "bf16 MFMA with fp32 residual accumulate"): the W x W hidden layers of the network are walked by THREE
// kernels per point chunk instead of three launches per layer (wide_kernel.h, which stays the fp32 path):
//
//   k_chain_fwd8   per 16-point tile: a_1 -> a_2 -> ... -> a_L through all hidden layers; the jet (value + k
//                  tangents) lives in registers as the NEXT layer's MFMA B operand, the weights stream through an
//                  LDS ring shared by the workgroup (one LDS-DMA copy per workgroup instead of one L2 read per wave);
//                  every a_l is written once for the reverse sweep.  Eight waves, two per SIMD, a tile per wave pair.
//   k_chain_bwd    the reverse chain: abar_L -> zbar_{L-1} -> abar_{L-1} -> ... -> abar_1 with the adjoint kept in
//                  the fp32 accumulators between layers (never rounded to bf16, never through HBM); reads each
//                  a_l once, writes each zbar_l once.  Four waves, a tile per wave.
//   k_chain_wgrad8 dW_l = sum_points zbar_l (x) a_l for ALL hidden layers in one launch: a workgroup owns one
//                  layer's whole 256 x 256 gradient in registers for its slice of the points (dW-stationary),
//                  streams zbar_l / a_l tiles through an LDS ring by LDS-DMA and takes the MFMA operands
//                  out of it with ds_read_b64_tr_b16 (the hardware transpose: contraction over points).  Eight waves.
//
//   k_chain_first_bwd / k_chain_last_fwd / k_chain_last_bwd: the thin first (d_in <= 3) and last (d_out <= 4) layers as
//                  streaming kernels on the same jets (the first layer's forward is part of k_chain_fwd8).
//
// HBM traffic per point and hidden layer: 2 KB (a written) + 2 + 2 (a, zbar in the reverse chain) + 4 (weight
// gradient) = 10 KB, against 16 KB for one launch per layer; weights come from L2 once per workgroup and layer.
//
// MFMA: v_mfma_f32_16x16x32_bf16 (the full-rate gfx950 form: 8 bf16 per lane and operand).  "acc layout =
// operand layout" carries over from fused_kernel.h with K = 32: lane (p = lane&15 point, q = lane>>4) holds, for
// k-step s, the 8 values j = 4h + r  <->  unit 32s + 16h + 4q + r, i.e. register r of accumulator tiles
// MT = 2s + h — two accumulator tiles, converted pairwise to bf16, ARE the B operand of k-step s.  The weights are
// packed to match (k index permuted inside each k-step, pinn_chain.hip k_chain_pack).
//
// Precision: bf16 operands, fp32 accumulate.  Jets (a_l, zbar_l) carry 8 significant bits.  The WEIGHTS are
// split hi + lo (two bf16, two MFMAs): rounding the weights themselves to bf16 moves this loss by 1.1e-2 and its
// gradient by 1.2e-1 (measured on the reference's 12 x 256 golden, tests/test_config3_gpu.py — a PINN's loss
// surface is stiff), the jets' rounding by 1.8e-3 / 3.8e-3.  The second MFMA doubles the matrix-pipe work of the two
// chain kernels; neither is bound by it (they are bound by what happens between their MFMAs).
//
// Jet storage ("chain layout", bf16): block (tile t, quantity c, k-step s) = 1 KB = [point p (16)][32 units], the
// 32 units ordered q, h, r — lane (p, q)'s 16 bytes are its B operand of k-step s, at byte (4p + q) * 16.  One
// wave-level 16-byte load or store moves exactly one contiguous block; a (point, 4 consecutive units) group is
// 8 contiguous bytes, which is what ds_read_b64_tr_b16 gathers.
#pragma once
#include "wide_kernel.h"

namespace pinn {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short sh4 __attribute__((ext_vector_type(4)));

constexpr int CHAIN_WAVES = 4;
constexpr int CHAIN_THREADS = CHAIN_WAVES * 64;
constexpr int CHAIN_RING = 6;              // weight slabs in the LDS ring of the reverse chain (next to its prefetch areas)
#ifndef PINN_CHAIN_RING_FWD
#define PINN_CHAIN_RING_FWD 6
#endif
constexpr int CHAIN_RING_FWD = PINN_CHAIN_RING_FWD;   // ... of the forward chain
// dynamic LDS of k_chain_fwd8 at padded width 16*ntw: ring + one spare slot (the ring's dummy copies) + every layer's
// bias + (folded first layer) 16 B of W_0 per unit.  ONE formula for the launcher and for wide_supports(): the depth
// limit of bf16 mode is whatever still fits the CU's 160 KB (L <= 43 at width 256, far more at width 128).
constexpr size_t CHAIN_LDS_LIMIT = 160 * 1024;
__host__ __device__ constexpr size_t chain_fwd8_lds_bytes(int ntw, int L, bool fold_first) {
  return (size_t)CHAIN_RING_FWD * (ntw / 2) * 2 * 1024 + (size_t)ntw * 1024 + (size_t)(L + 1) * 16 * ntw * 4 +
         (fold_first ? (size_t)16 * ntw * 16 : 0);
}
#ifndef PINN_BWD_PREFETCH
#define PINN_BWD_PREFETCH 8
#endif
constexpr int BWD_PREFETCH = PINN_BWD_PREFETCH;       // k_chain_bwd: the same, inside a layer
#ifndef PINN_FWD8_PREFETCH
#define PINN_FWD8_PREFETCH 4
#endif
constexpr int FWD8_PREFETCH = PINN_FWD8_PREFETCH;     // k-steps of the next slab read before the barrier (0: off)
// Weight precision of the REVERSE chain.  Measured on the reference's 12 x 256 golden (G10): rounding the weights to
// bf16 in the FORWARD pass moves the gradient by 1.2e-1 (the loss is evaluated at a shifted point of a stiff
// surface) — the forward chain always multiplies by hi + lo.  Rounding them in the reverse pass alone is a random,
// averaging error: gradient 5.2e-3 instead of 3.3e-3 for 7 % of the step (2^20 points: 40.0 vs 42.9 ms).  Default 1
// (hi + lo in both directions): the stated bf16 tolerance of this engine is 5e-3.
#ifndef PINN_CHAIN_BWD_LO
#define PINN_CHAIN_BWD_LO 1
#endif
constexpr int WG_UNITS = 4;                // LDS ring slots of the weight-gradient kernel (half tiles)

struct ChainParams {
  int L;                      // hidden layers of the network (L - 1 hidden W x W matrices: layers 1 .. L-1)
  int64_t n_tiles;            // tiles in this chunk
  int64_t jet_stride;         // elements (bf16) between consecutive layers' jets
  int64_t w_plane;            // bytes between the 1 KB pieces of one weight slab (see k_chain_pack: pieces are spread
                              // over memory so that the CUs of an XCD, reading the same slab together, load many L2 channels)
  const unsigned short* Wf;   // packed weight fragments, layers 1 .. L-1 (fwd) — see k_chain_pack
  const unsigned short* WTf;  // packed transposed weight fragments (bwd)
  const float* bias;          // padded biases, fp32: layer l at bias + l * WP
  unsigned short* A;          // a_1 at A, a_2 at A + jet_stride, ... a_L
  unsigned short* Z;          // zbar_1 at Z, ... zbar_{L-1}
  unsigned short* GL;         // abar_L (input of the reverse chain)
  unsigned short* G1;         // abar_1 (its output)
  float* dW;                  // flat torch-layout gradient (wgrad)
  int spill;                  // fwd: 0 = forward only (no a_l stores except a_L)
  int W;                      // real hidden width
  int n_slices;               // wgrad: point slices per layer
  int64_t w_off1, w_per;      // flat offsets: W_1 at w_off1, W_{l+1} - W_l = w_per; b_l at w_off(l) + W*W
  unsigned long long* diag;   // -DPINN_CHAIN_DIAG builds only: per-phase cycle sums of workgroup 0 / wave 0
  // first layer folded into k_chain_fwd8 (d_in <= 3): a_1 is computed from the coordinates, never loaded
  const float* X;             // (n_points, d_in) row-major, the WHOLE point set
  const float* W0;            // padded first-layer weights, row-major [WP][16] fp32 (wide engine's k_wide_pack)
  int64_t n_points, tile0;    // points in X; index of this chunk's first tile
  int d_in;
  int dir_col[3];             // input column of tangent 1, 2, 3
  // last layer's reverse pass in one kernel (k_chain_last_bwd, d_out <= 4)
  const float* gout;          // G: [tile][quantity][256] fp32, lane (p, q) x 4: outputs 4q .. 4q+3 of point p (k_wide_fwd)
  const float* WLT;           // padded transposed last-layer weights [WP][16] fp32
  float* dWL;                 // flat gradient of W_L (d_out, W) row-major, then b_L
  int d_out;
};

// Diagnostic build (-DPINN_CHAIN_DIAG, tools/chain_diag.sh): s_memtime stamps per phase, summed by one wave.
// Stamps serialise the instruction stream: read the SHARES, not the total.
#ifdef PINN_CHAIN_DIAG
#define CHAIN_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); dg[i] += t_ - tprev; tprev = t_; } while (0)
#define CHAIN_DIAG_BEGIN unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter()
#define CHAIN_DIAG_END(P) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (P).diag) for (int i_ = 0; i_ < 8; ++i_) (P).diag[i_] = dg[i_]; } while (0)
// 8-wave kernels: wave 0 (an early wave) into the kernel's slots, wave 4 (its late SIMD partner) into slots 24..31
#define CHAIN_DIAG_END8(P) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (P).diag) for (int i_ = 0; i_ < 8; ++i_) (P).diag[(threadIdx.x ? 24 : 0) + i_] = dg[i_]; } while (0)
#else
#define CHAIN_DIAG_END8(P) do { } while (0)
#define CHAIN_STAMP(i) do { } while (0)
#define CHAIN_DIAG_BEGIN do { } while (0)
#define CHAIN_DIAG_END(P) do { } while (0)
#endif

#ifdef PINN_CHAIN_EXP_NOMFMA   // timing experiment only (results are garbage): how fast do the weight copies run alone?
__device__ __forceinline__ f4 mfma32(bf8 a, bf8 b, f4 c) { asm volatile("" ::"v"(a), "v"(b)); return c; }
#else
__device__ __forceinline__ f4 mfma32(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 1 KB LDS-DMA copy: lane i moves 16 bytes from src + 16 i to lds_dst + 16 i (lds_dst wave-uniform)
template <int AUX = 0>
__device__ __forceinline__ void dma_1k(const void* src, void* lds_dst, int lane) {
  __builtin_amdgcn_global_load_lds(
      (const void __attribute__((address_space(1)))*)((const char*)src + (unsigned)lane * 16u),
      (void __attribute__((address_space(3)))*)lds_dst, 16, 0, AUX);
}

__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }

// Jet block access through BUFFER instructions: a scalar resource (base = this wave's tile inside one jet, range =
// the tile's K1 * NS KB: out-of-range accesses are dropped by the hardware), ONE 32-bit per-lane byte offset and a
// scalar block offset.  With plain pointers the compiler builds a 64-bit VGPR address per block, hoists them out
// of the loops and spills them by the hundred (506 spilled registers in k_chain_bwd at K1 = 4).
typedef unsigned u4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t jet_rsrc(const void* tile_base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile_base), 0, bytes, 0x00020000);
}
// Cache policy of the jet streams.  The chain kernels re-read the SAME packed weights (2.8 MB per direction at
// 12 x 256: they fit an XCD's 4 MB L2) once per tile batch, while ~50 MB of jets per XCD pass through that L2 in
// between: with default-policy jet traffic the weight copies ran at the beyond-L2 LDS-DMA rate (~25 GB/s per CU,
// = the 16 KB of every GEMM step: the steps were bound by them, not by their 64 MFMAs).  Jets are written once
// and read once much later: stores go out write-through / no L2 allocation (sc1), loads are non-temporal (nt).
#ifndef PINN_CHAIN_JET_ST_AUX
#define PINN_CHAIN_JET_ST_AUX 18   // sc1 | nt (measured: reverse chain 15.05 -> 14.8 ms against sc1 alone; default policy: forward +0.35 ms)
#endif
#ifndef PINN_CHAIN_JET_LD_AUX
#define PINN_CHAIN_JET_LD_AUX 2    // nt
#endif
__device__ __forceinline__ bf8 ld_blk(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int blk_off) {
  return __builtin_bit_cast(bf8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, blk_off, PINN_CHAIN_JET_LD_AUX));
}
__device__ __forceinline__ void st_blk(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int blk_off, bf8 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)lane_off, blk_off, PINN_CHAIN_JET_ST_AUX);
}

// a wave-uniform 64-bit value the compiler cannot prove uniform (derived from threadIdx): moved to SGPRs so
// that jet accesses become  scalar base + one 32-bit lane offset + immediate  instead of 64-bit VGPR addresses
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// tanh for values that are rounded to 8 significant bits right afterwards: 1 - 2 / (exp(2x) + 1), absolute
// error ~1e-7 (v_exp_f32 + v_rcp_f32); the polynomial branch of residuals.h::tanh_f32 buys nothing here.
__device__ __forceinline__ float tanh_bf(float x) {
  const float e = __expf(2.f * x);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}


// The weight ring of the two chain kernels.  Slabs are consumed in a fixed order (tile batch, layer, output tile);
// `next` walks that order one slab ahead of the R - 1 in flight.  Everything is wave-uniform.
template <int NTW, int SLAB, int R, bool DESCENDING, int NW = CHAIN_WAVES>
struct SlabRing {
  const char* base; char* lds; int nh, wave, lane; int64_t plane;
  int64_t issued, total;      // slabs issued so far / slabs this workgroup consumes
  int li, MT, slot;           // of the next slab to issue
  int cslot;                  // ring slot of the slab being consumed
  __device__ __forceinline__ void init(const void* w, int64_t plane_, char* smem, int nh_, int64_t total_, int wave_, int lane_) {
    base = (const char*)w; plane = plane_; lds = smem; nh = nh_; total = total_; wave = wave_; lane = lane_;
    issued = 0; li = DESCENDING ? nh - 1 : 0; MT = 0; slot = 0; cslot = 0;
  }
  __device__ __forceinline__ void issue() {
    if (issued >= total) return;
    constexpr int QDMA = SLAB / NW / 1024;
    static_assert(QDMA * NW * 1024 == SLAB, "every wave copies whole 1 KB pieces of a slab");
    // piece i of slab n lives at  i * plane + n * 1 KB
    const char* src = base + (int64_t)(wave * QDMA) * plane + ((int64_t)(li * NTW + MT)) * 1024;
    char* dst = lds + slot * SLAB + wave * (SLAB / NW);
#pragma unroll
    for (int i = 0; i < QDMA; ++i) dma_1k(src + i * plane, dst + i * 1024, lane);
    ++issued;
    slot = slot + 1 == R ? 0 : slot + 1;
    if (++MT == NTW) { MT = 0; li = DESCENDING ? (li == 0 ? nh - 1 : li - 1) : (li + 1 == nh ? 0 : li + 1); }
  }
  // Before consuming slab number `g` (0-based): its copies (this wave's quarter) have landed once at most the
  // copies of the R - 2 younger slabs — plus the EXTRA jet stores this wave is known to have issued after
  // them (`extra_issued`; vmcnt retires in issue order, so every younger operation must be counted or the wait
  // is for too much) — are outstanding.  Near the end of the sequence fewer younger slabs exist: drain everything.
  template <int EXTRA, int RB = R>     // RB = R - 1: the target is the NEXT step's slab (one slab less in flight)
  __device__ __forceinline__ void wait_landed(int64_t g, bool extra_issued) {
    constexpr int QDMA = SLAB / NW / 1024;
    constexpr int N = (RB - 2) * QDMA;
    if (g + R - 1 > total) wait_vm<0>();           // (issue() has been skipping: fewer than R - 2 younger slabs)
    else if (EXTRA > 0 && N + EXTRA <= 63 && extra_issued) wait_vm<(N + EXTRA <= 63 ? N + EXTRA : N)>();
    else wait_vm<N>();
  }
  __device__ __forceinline__ const char* consume_ptr() const { return lds + cslot * SLAB + lane * 16; }
  __device__ __forceinline__ void consumed() { cslot = cslot + 1 == R ? 0 : cslot + 1; }
};

// compile-time loop: body(std::integral_constant<int, I>) for I = 0 .. N-1 (the step index parameterises vmcnt immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& body) {
  if constexpr (I < N) {
    body(std::integral_constant<int, I>{});
    static_for<I + 1, N>(body);
  }
}

// Jet stores are SPREAD over the GEMM steps that follow the phase producing them (two 1 KB blocks per step until
// the K1 * NS blocks are out): the registers they come from are the next GEMM's B operand and stay live anyway,
// and a burst of K1 * NS KB per wave at the end of the activation phase ran at the CU's store rate (~10 B/clk:
// 38 % of the forward kernel's cycles, tools/chain_diag.sh) while the matrix pipe idled.
template <int K1, int NS>
__host__ __device__ constexpr int chain_stores_at(int step) {   // blocks stored at GEMM step `step` (< 0: none)
  return step < 0 ? 0 : (K1 * NS - 2 * step >= 2 ? 2 : (K1 * NS - 2 * step > 0 ? K1 * NS - 2 * step : 0));
}
// stores younger than slab g's copies when step MT of a layer starts: those of the R - 2 previous steps and of the
// step that issued slab g (its stores follow its copies).  CUR / PREV: are stores riding on this / the previous
// layer's steps?
template <int K1, int NS, int NTW, int R, int MT, bool CUR, bool PREV>
__host__ __device__ constexpr int chain_younger_stores() {
  int n = 0;
  for (int j = 1; j <= R - 1; ++j) {
    const int st = MT - j;
    if (st >= 0) n += CUR ? chain_stores_at<K1, NS>(st) : 0;
    else n += PREV ? chain_stores_at<K1, NS>(NTW + st) : 0;
  }
  return n;
}

// vector-memory operations a wave issues at GEMM steps [m0, m1) of the reverse chain besides prefetch copies
template <int K1, int NS, int QD>
__host__ __device__ constexpr int chain_ops_in_steps(int m0, int m1) {
  int n = 0;
  for (int m = m0; m < m1; ++m) n += QD + chain_stores_at<K1, NS>(m);
  return n;
}
// prefetch copies (two per step from step 0 until the NPF copies are out — early, so that they are old by the time
// the next phase waits for them) issued at step m / among the R - 1 steps before step MT of the same layer
template <int NPF>
__host__ __device__ constexpr int chain_pf_at(int m) {
  return m < 0 ? 0 : (NPF - 2 * m >= 2 ? 2 : (NPF - 2 * m > 0 ? NPF - 2 * m : 0));
}
template <int R, int NPF>
__host__ __device__ constexpr int chain_younger_prefetch(int MT) {
  int n = 0;
  for (int j = 1; j <= R - 1; ++j) n += chain_pf_at<NPF>(MT - j);
  return n;
}
__host__ __device__ constexpr int chain_pf_pieces(int NTW, int K1) {   // k-step pieces of a_l prefetched into LDS
  return NTW / 2 < NTW / K1 ? NTW / 2 : NTW / K1;
}

// ------------------------------------------------------------------------------------------------------------
// Forward chain.  Weight slab (layer l, output tile MT) = [hi | lo][k-step s][lane][8 bf16]: NS * 2 KB.
// ============================================================================================================
// Two waves per SIMD ("p8" kernels).  tools/ubench_ldsdma.hip: with ONE wave per SIMD a GEMM step of the chain
// (64 MFMAs = 1024 matrix-pipe cycles) takes ~1800 cycles — bringing the 16 KB weight slab into LDS blocks the
// waves for ~400 cycles (the CU moves ~40 B/clk whichever way the bytes are staged, and all four waves wait
// on it at once), the slab's 16 fragment reads for ~260, the barrier for ~130 — and nothing overlaps any of it.
// So the chain kernels run EIGHT waves per workgroup, two per SIMD, each with half the registers: a wave owns
// 8 points (one half of a 16-point tile) x all quantities.  The 16 MFMA columns of a wave carry 8 points x 2
// quantities: column = 8 * par + p8, group G holds quantities 2G (par 0) and 2G + 1 (par 1), so the jet of K1 = 4
// is two column groups — 128 accumulator registers and 64 of B operand per wave instead of 256 + 128.  The 16-point
// tile, its blocks in memory and the weight ring are unchanged (a workgroup is still four tiles = 64 points per
// pass over the weights); the two waves of a tile are independent of each other.  Everything a point needs across
// its quantities (tanh' for the tangents, the cross term of the adjoint) sits in the two lanes p8 and p8 + 8 of
// the same wave: one DPP row rotate.
constexpr int P8_WAVES = 8;
constexpr int P8_THREADS = P8_WAVES * 64;

// two stores (each a pair of quantity blocks) per step until the NGS = NG * NS stores of a layer are out: all in the
// first half of the layer, so that they are old — retired, or nearly — when the batch boundary drains the queue
template <int NGS>
__host__ __device__ constexpr int p8_stores_at(int step) { return step >= 0 && 2 * step < NGS ? 2 : 0; }
template <int NGS, int NTW, int R, int MT, bool CUR, bool PREV>
__host__ __device__ constexpr int p8_younger_stores() {
  int n = 0;
  for (int j = 1; j <= R - 1; ++j) {
    const int st = MT - j;
    if (st >= 0) n += CUR ? p8_stores_at<NGS>(st) : 0;
    else n += PREV ? p8_stores_at<NGS>(NTW + st) : 0;
  }
  return n;
}

// lanes 8..15 of every row of 16 take the value of lane - 8; lanes 0..7 keep theirs (bank mask 0b1100)
__device__ __forceinline__ float dpp_hi_from_lo(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x128, 0xf, 0xc, false));
}

// Two units (r, r + 1) at a time: the arithmetic is packed fp32 (v_pk_mul / v_pk_fma), selections by lane are
// multiplications with parf = (float)par — the vector ALU's issue port is shared with the MFMAs of the SIMD's other wave.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 dpp2_hi_from_lo(f2 v) { return f2{dpp_hi_from_lo(v[0]), dpp_hi_from_lo(v[1])}; }
// Roles inside a step.  Waves w and w + 4 share a SIMD (dispatch order); run in lockstep they would both stall on
// their weight copies, then both on their fragment reads, then fight over the matrix pipe.  So the two halves of
// the workgroup order a step differently: waves 0-3 ("early") issue their copies and stores and run the
// activation of the PREVIOUS output tile first, then the step's MFMAs; waves 4-7 ("late") start with the MFMAs
// and issue copies / stores / the activation of THIS output tile after them.  An output tile is final after its
// own step (the loop is output-stationary: step MT multiplies the 16 x W slab of output tile MT with the whole
// input jet), so the activation — the only vector-ALU-heavy part — always has a partner's MFMAs to hide behind,
// only one or two accumulator tiles are ever live, and the next layer's operand is built piece by piece (`bn`).
// FOLD (d_in <= 3, every network of the reference): the first layer is part of this kernel.  z = W_0 x + b_0 is three
// FMAs per unit and its tangents are columns of W_0, so a tile's a_1 is computed in registers from 12 bytes per point
// instead of being written by one kernel (2 KB per point) and read back by this one.
template <int NTW, int K1, bool FOLD>
__global__ __launch_bounds__(P8_THREADS, 2) void k_chain_fwd8(const ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = NTW / 2;
  constexpr int NG = (K1 + 1) / 2;
  constexpr int SLAB = NS * 2 * 1024;
  constexpr int QD = SLAB / P8_WAVES / 1024;
  constexpr int R = CHAIN_RING_FWD;
  constexpr int NST = NG * NS;                   // stores per wave and layer
  static_assert(NST <= 2 * NTW && NST % 2 == 0, "two stores per step");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool early = wave < 4;
  const int q = lane >> 4, col = lane & 15, par = col >> 3, p8 = col & 7, half = wave & 1;
  // byte offset of this lane inside the tile's jet: row 8 half + p8 of the block of quantity `par` (+ 2G per group)
  const unsigned lvo = (unsigned)par * (NS * 1024) + (4u * (8u * half + p8) + q) * 16u;
  // odd K1: the last group's par-1 lanes have no quantity — pushed out of the buffer's range (loads 0, stores dropped;
  // only the per-lane offset is range-checked, the scalar block offset is not)
  const unsigned lvo_last = ((K1 & 1) && par) ? 0x40000000u : lvo;
  const float parf = (float)par;
  const int nh = P.L - 1;
  const int64_t n_tb = (P.n_tiles + CHAIN_WAVES - 1) / CHAIN_WAVES;
  const int64_t my_tb = (n_tb - blockIdx.x + gridDim.x - 1) / gridDim.x;
  float* bias_lds = reinterpret_cast<float*>(smem + (R + 1) * SLAB);
  for (int i = threadIdx.x; i < (P.L + 1) * 16 * NTW; i += P8_THREADS) bias_lds[i] = P.bias[i];
  f4* w0_lds = reinterpret_cast<f4*>(bias_lds + (P.L + 1) * 16 * NTW);      // FOLD: (W_0[u][0..2], 0) per unit
  if constexpr (FOLD) {
    for (int i = threadIdx.x; i < 16 * NTW; i += P8_THREADS) w0_lds[i] = f4{P.W0[i * 16], P.W0[i * 16 + 1], P.W0[i * 16 + 2], 0.f};
  }
  __syncthreads();
  // The ring.  A wave's vector-memory operations retire in issue order, so "slab g has landed" is a vmcnt wait for
  // at most the operations issued after its copies.  To make that number a compile-time constant of the step, every
  // step issues the same operations whatever the layer: QD copies (past the last slab: of slab 0 into a spare slot
  // nobody reads) and, for steps < NST, one jet store (layers with nothing to store use an empty buffer range:
  // the store is dropped, and still counted).
  SlabRing<NTW, SLAB, R, false, P8_WAVES> ring;
  ring.init(P.Wf, P.w_plane, smem, nh, my_tb * nh * NTW, wave, lane);
  auto issue = [&]() {
    if (ring.issued < ring.total) ring.issue();
    else {
#pragma unroll
      for (int i = 0; i < QD; ++i)
        dma_1k((const char*)P.Wf + (int64_t)(wave * QD + i) * P.w_plane, smem + R * SLAB + (wave * QD + i) * 1024, lane);
    }
  };
  for (int g0 = 0; g0 < R - 1; ++g0) issue();
  // Fragment prefetch across the barrier.  The barrier of step g certifies slab g + 1 (not g), one slab less in flight:
  // a wave can then read the first PFN k-steps of the NEXT slab right after its own MFMAs, and opens its next step on
  // fragments that are already in registers instead of an LDS round trip behind the barrier.
  constexpr int PFN = FWD8_PREFETCH;
  bf8 pfh[PFN > 0 ? PFN : 1], pfl[PFN > 0 ? PFN : 1];
  auto prefetch = [&]() {
    if constexpr (PFN > 0) {
      const char* sl = ring.consume_ptr();
#pragma unroll
      for (int s = 0; s < PFN; ++s) {
        pfh[s] = *reinterpret_cast<const bf8*>(sl + s * 1024);
        pfl[s] = *reinterpret_cast<const bf8*>(sl + (NS + s) * 1024);
      }
    }
  };
  if constexpr (PFN > 0) {
    wait_vm<(R - 2) * QD>();            // slab 0 of the R - 1 just requested
    __builtin_amdgcn_s_barrier();
    prefetch();
  }
  CHAIN_DIAG_BEGIN;
  constexpr int TILE_BYTES = K1 * NS * 1024;
  bf8 bj[NG][NS], bn[NG][NS], bx[NG][NS];
#pragma unroll
  for (int G = 0; G < NG; ++G)
#pragma unroll
    for (int s = 0; s < NS; ++s) bn[G][s] = bf8{0, 0, 0, 0, 0, 0, 0, 0};
  // Tile batches.  A batch's a_1 is loaded into `bx` one layer ahead (at the start of the previous batch's last
  // layer), its a_L is stored from `bn` at the start of the next batch: loads and stores retire in issue order, and a
  // load issued right behind 16 KB of stores per wave would wait for all of them to reach HBM.
  auto tile_of = [&](int64_t tb_, bool& live_) {
    int64_t t_ = tb_ * CHAIN_WAVES + (wave >> 1);
    live_ = tb_ < n_tb && t_ < P.n_tiles;
    if (t_ >= P.n_tiles) t_ = P.n_tiles - 1;
    return uniform64(t_) * (K1 * NS * 512);
  };
  auto load_a1 = [&](int64_t tbase_, bool any) {
    const __amdgpu_buffer_rsrc_t a1r = jet_rsrc(P.A + tbase_, any ? TILE_BYTES : 0);
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
      for (int s = 0; s < NS; ++s) bx[G][s] = ld_blk(a1r, G == NG - 1 ? lvo_last : lvo, (2 * G * NS + s) * 1024);
  };
  auto store_aL = [&](int64_t tbase_, bool live_) {     // dead waves / before the first batch: dropped, counted
    const __amdgpu_buffer_rsrc_t dstL = jet_rsrc(P.A + (int64_t)nh * P.jet_stride + tbase_, live_ ? TILE_BYTES : 0);
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
      for (int s = 0; s < NS; ++s) st_blk(dstL, G == NG - 1 ? lvo_last : lvo, (2 * G * NS + s) * 1024, bn[G][s]);
  };
  // FOLD: this lane's point of batch tb_ -> its coordinates (three loads whatever d_in: missing columns and points
  // beyond the set read as 0 from an out-of-range offset), then the tile's a_1 from them
  float xr[3] = {0.f, 0.f, 0.f};
  // (the resource covers THIS CHUNK's rows of X: buffer offsets are 32 bits, the whole point set may pass 2 GB)
  const int64_t x_rest = (P.n_points - P.tile0 * 16) * P.d_in * 4;
  const __amdgpu_buffer_rsrc_t xrs = jet_rsrc(P.X + P.tile0 * 16 * P.d_in, (int)(x_rest < 0x7fffffff ? (x_rest > 0 ? x_rest : 0) : 0x7fffffff));
  auto load_x = [&](int64_t tb_) {
    int64_t t_ = tb_ * CHAIN_WAVES + (wave >> 1);
    if (t_ >= P.n_tiles) t_ = P.n_tiles - 1;
    const int64_t ptl = t_ * 16 + 8 * half + p8;             // row inside this chunk
    const bool ok = tb_ < n_tb && P.tile0 * 16 + ptl < P.n_points;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned off = ok && j < P.d_in ? (unsigned)(ptl * P.d_in + j) * 4u : 0x7ffffff0u;
      xr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)off, 0, 0));
    }
  };
  auto make_a1 = [&]() {
    // tangent c (quantity 2G + par >= 1) of the input is the unit vector of column dir_col[c - 1]: z-dot = W_0[u][that column]
    const int cA = P.dir_col[0], cB = P.dir_col[1], cC = P.dir_col[2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int u = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
        const f4 w = w0_lds[u];
        const float z = fmaf(w[2], xr[2], fmaf(w[1], xr[1], fmaf(w[0], xr[0], bias_lds[u])));
        const float av = tanh_bf(z), sv = fmaf(-av, av, 1.f);
        auto col = [&](int cc) { return cc == 0 ? w[0] : (cc == 1 ? w[1] : w[2]); };
        const float t1 = sv * col(cA), t2 = sv * col(cB), t3 = sv * col(cC);
        bj[0][s][j] = (__bf16)(par ? (K1 > 1 ? t1 : 0.f) : av);
        if constexpr (NG > 1) bj[1][s][j] = (__bf16)(par ? (K1 > 3 ? t3 : 0.f) : t2);
      }
  };
  bool live = false, prev_live = false, next_live = false;
  int64_t tbase = tile_of(blockIdx.x, live), prev_base = 0, next_base = 0;
  if constexpr (FOLD) load_x(blockIdx.x); else load_a1(tbase, true);
  for (int64_t tb = blockIdx.x; tb < n_tb; tb += gridDim.x) {
    if constexpr (FOLD) make_a1();
    else {
#pragma unroll
      for (int G = 0; G < NG; ++G)
#pragma unroll
        for (int s = 0; s < NS; ++s) bj[G][s] = bx[G][s];
    }
    store_aL(prev_base, prev_live);
    CHAIN_STAMP(6);
    for (int l = 1; l <= nh; ++l) {
      f4 accp[NG];
      // a_l goes out during this layer's steps (a_1 only if this kernel made it: otherwise it is in memory already)
      const __amdgpu_buffer_rsrc_t dst = jet_rsrc(P.A + (int64_t)(l - 1) * P.jet_stride + tbase, live && P.spill && (l >= 2 || FOLD) ? TILE_BYTES : 0);
      const float* bl = bias_lds + l * (16 * NTW);
      if (l == nh) {
        next_base = tile_of(tb + gridDim.x, next_live);
        if constexpr (FOLD) load_x(tb + gridDim.x); else load_a1(next_base, tb + gridDim.x < n_tb);
      }
      // activation of output tile M: the value lanes (group 0, par 0) take tanh(z + b); every other column of the
      // point is a tangent and is scaled by the point's 1 - a^2 (one row rotate away for the par-1 lanes)
      auto act = [&](auto m_, const f4 (&a)[NG]) {
        constexpr int M = decltype(m_)::value, s = M / 2, h = M % 2;
        const f4 b4 = *reinterpret_cast<const f4*>(bl + 16 * M + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {                        // two units at a time, packed fp32
          const f2 x = f2{a[0][r], a[0][r + 1]};
          const f2 e2 = (x + f2{b4[r], b4[r + 1]}) * 2.885390081777927f;      // 2 log2(e) (x + b)
          const f2 ex = f2{__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])} + 1.f;
          const f2 rc = f2{__builtin_amdgcn_rcpf(ex[0]), __builtin_amdgcn_rcpf(ex[1])};
          const f2 av = 1.f - 2.f * rc;                         // tanh(x + b): meaningful on the par-0 lanes
          const f2 sv = dpp2_hi_from_lo(1.f - av * av);         // the point's 1 - a^2 on both of its lanes
          const f2 o0 = (x * sv - av) * parf + av;              // par 0: the value; par 1: tangent 1
          bn[0][s][4 * h + r] = (__bf16)o0[0]; bn[0][s][4 * h + r + 1] = (__bf16)o0[1];
#pragma unroll
          for (int G = 1; G < NG; ++G) {
            const f2 og = f2{a[G][r], a[G][r + 1]} * sv;
            bn[G][s][4 * h + r] = (__bf16)og[0]; bn[G][s][4 * h + r + 1] = (__bf16)og[1];
          }
        }
      };
      auto side = [&](auto m_) {      // this step's copies and its share of the a_l stores
        constexpr int M = decltype(m_)::value;
        issue();
        if constexpr (2 * M < NST) {
#pragma unroll
          for (int i = 2 * M; i < 2 * M + 2; ++i)
            st_blk(dst, i / NS == NG - 1 ? lvo_last : lvo, (2 * (i / NS) * NS + i % NS) * 1024, bj[i / NS][i % NS]);
        }
      };
      static_for<0, NTW>([&](auto mt_) {
        constexpr int MT = decltype(mt_)::value;
        CHAIN_STAMP(5);
        {
          // younger than this step's slab: the copies of R - 2 slabs, the stores of the last R - 1 steps and, in
          // the first R - 1 steps of a layer, what was issued at its start: the a_L stores of the previous batch
          // (first layer), the a_1 loads of the next one (last layer)
          // (with the fragment prefetch the target is the NEXT step's slab, issued R - 2 steps ago)
          constexpr int RB = PFN > 0 ? R - 1 : R;
          constexpr int N = (RB - 2) * QD + p8_younger_stores<NST, NTW, RB, MT, true, true>();
          constexpr int NLD = FOLD ? 3 : NST;          // loads issued at the start of the last layer
          static_assert(N + NST + NLD <= 63, "vmcnt range");
          if (MT < RB - 1 && (l == 1 || l == nh)) {
            if (l == 1 && l == nh) wait_vm<N + NST + NLD>();
            else if (l == 1) wait_vm<N + NST>();
            else wait_vm<N + NLD>();
          } else wait_vm<N>();
        }
        CHAIN_STAMP(0);
        __builtin_amdgcn_s_barrier();
        CHAIN_STAMP(1);
        if (early) {
          side(mt_);
          CHAIN_STAMP(4);
          if constexpr (MT > 0) act(std::integral_constant<int, (MT > 0 ? MT - 1 : 0)>{}, accp);
          CHAIN_STAMP(3);
        }
        f4 accc[NG];
#pragma unroll
        for (int G = 0; G < NG; ++G) accc[G] = f4{0.f, 0.f, 0.f, 0.f};
        const char* sl = ring.consume_ptr();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bf8 ahi = s < PFN ? pfh[s < PFN ? s : 0] : *reinterpret_cast<const bf8*>(sl + s * 1024);
          const bf8 alo = s < PFN ? pfl[s < PFN ? s : 0] : *reinterpret_cast<const bf8*>(sl + (NS + s) * 1024);
#pragma unroll
          for (int G = 0; G < NG; ++G) {
            accc[G] = mfma32(ahi, bj[G][s], accc[G]);
            accc[G] = mfma32(alo, bj[G][s], accc[G]);
          }
        }
        ring.consumed();
        prefetch();
        CHAIN_STAMP(2);
        if (!early) {
          side(mt_);
          CHAIN_STAMP(4);
          act(mt_, accc);
          CHAIN_STAMP(3);
        }
#pragma unroll
        for (int G = 0; G < NG; ++G) accp[G] = accc[G];
      });
      CHAIN_STAMP(5);
      if (early) act(std::integral_constant<int, NTW - 1>{}, accp);
      CHAIN_STAMP(3);
      if (l < nh) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
#pragma unroll
          for (int s = 0; s < NS; ++s) bj[G][s] = bn[G][s];
      }
      CHAIN_STAMP(6);
    }
    prev_base = tbase; prev_live = live;
    tbase = next_base; live = next_live;
  }
  store_aL(prev_base, prev_live);
  wait_vm<0>();
  CHAIN_DIAG_END8(P);
}

// ------------------------------------------------------------------------------------------------------------
// Reverse chain.  State between layers: abar_{l+1} in the fp32 accumulators.  Per hidden matrix l (L-1 .. 1):
//   zbar_l = activation adjoint(abar_{l+1}, a_{l+1})   (fused_kernel.h activate_adjoint: tanh'' term included)
//   store zbar_l (weight-gradient operand);  abar_l = W_l^T zbar_l  (same chain on the packed W^T)
template <int NTW, int K1>
__global__ __launch_bounds__(CHAIN_THREADS, 1) void k_chain_bwd(const ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = NTW / 2;
  constexpr bool LO = PINN_CHAIN_BWD_LO != 0;
  constexpr int SLAB = NS * (LO ? 2 : 1) * 1024;     // the hi pieces are the first NS planes of a packed slab
  constexpr int QD = SLAB / 4 / 1024;
  constexpr int R = CHAIN_RING;
  // fragment prefetch across the barrier (see k_chain_fwd8): inside a layer the barrier of a step certifies the NEXT
  // step's slab, whose first PFB k-steps are read right after this step's MFMAs
  constexpr int PFB = BWD_PREFETCH < NS ? BWD_PREFETCH : NS;   // (all of them at width 256: 15.0 -> 14.0 ms; 2: 14.4, 4: 14.2)
  constexpr int RB = PFB > 0 ? R - 1 : R;
  // the first PF k-step pieces of the NEXT adjoint phase's a_l are copied into LDS (two 1 KB LDS-DMA copies per GEMM
  // step from step 0) while this layer's GEMM runs: the phase then starts on data that is already on
  // chip and its remaining pieces stream into registers behind it.  (All of a_l does not fit: ring + 4 waves x 16 KB
  // = the CU's 160 KB at width 256.)  Requested in place the loads left an HBM round trip in front of every phase.
  constexpr int PF = chain_pf_pieces(NTW, K1);
  constexpr int PF_BYTES = PF * K1 * 1024;           // per wave (<= NTW KB)
  constexpr int NPF = PF * K1;                       // copies per layer, two per GEMM step from step 0
  constexpr int PF_STEPS = (NPF + 1) / 2;
  static_assert(PF_STEPS <= NTW, "prefetch copies fit the GEMM's steps");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4;
  const unsigned lpos = (4u * (lane & 15) + q) * 16u;
  const int nh = P.L - 1;
  const int64_t n_tb = (P.n_tiles + CHAIN_WAVES - 1) / CHAIN_WAVES;
  const int64_t my_tb = (n_tb - blockIdx.x + gridDim.x - 1) / gridDim.x;
  SlabRing<NTW, SLAB, R, true> ring;               // slab order: layers nh-1 .. 0 (descending), tiles 0 .. NTW-1
  ring.init(P.WTf, P.w_plane, smem, nh, my_tb * nh * NTW, wave, lane);
  char* pf = smem + R * SLAB + wave * PF_BYTES;    // this wave's prefetch area
  for (int g0 = 0; g0 < R - 1; ++g0) ring.issue();
  int64_t g = 0;
  CHAIN_DIAG_BEGIN;
  for (int64_t tb = blockIdx.x; tb < n_tb; tb += gridDim.x) {
    int64_t t = tb * CHAIN_WAVES + wave;
    const bool live = t < P.n_tiles;
    if (!live) t = P.n_tiles - 1;
    const int64_t tbase = uniform64(t) * (K1 * NS * 512);
    constexpr int TILE_BYTES = K1 * NS * 1024;
    const __amdgpu_buffer_rsrc_t glp = jet_rsrc(P.GL + tbase, TILE_BYTES);
    f4 acc[K1][NTW];
    for (int l = nh; l >= 1; --l) {
      // ---- activation adjoint on a_{l+1} (stored at A + l * jet_stride) -> zbar_l as the next B operand.
      // abar_{l+1} sits in the accumulators, except for l = nh: abar_L comes from the output layer's reverse
      // kernel (bf16, chain layout) and is read piece by piece next to a_L — preloading it into the accumulators
      // made the compiler keep a VGPR copy of all 64 K1 values beside the AGPR one (330 spilled registers).
      const __amdgpu_buffer_rsrc_t aop = jet_rsrc(P.A + (int64_t)l * P.jet_stride + tbase, TILE_BYTES);
      const __amdgpu_buffer_rsrc_t zdst = jet_rsrc(P.Z + (int64_t)(l - 1) * P.jet_stride + tbase, TILE_BYTES);
      bf8 zj[K1][NS];
      CHAIN_STAMP(2);
      // a_{l+1} arrives in k-step pieces: the first PF0 from this wave's LDS prefetch area (l < nh), the rest
      // requested AHEAD pieces before their use and no earlier (left alone the compiler hoists every load of the
      // phase to its top and spills)
      auto phase = [&](auto from_mem) {
        constexpr bool MEM = decltype(from_mem)::value;
        constexpr int AHEAD = MEM ? 1 : 2;         // (two streams in the first phase: half the look-ahead each)
        constexpr int PF0 = MEM ? 0 : PF;
        bf8 av[AHEAD + 1][K1], gv[MEM ? AHEAD + 1 : 1][K1];
        auto request = [&](int s2, int slot) {
#pragma unroll
          for (int c = 0; c < K1; ++c) {
            av[slot][c] = ld_blk(aop, lpos, (c * NS + s2) * 1024);
            if constexpr (MEM) gv[slot][c] = ld_blk(glp, lpos, (c * NS + s2) * 1024);
          }
        };
#pragma unroll
        for (int s0 = PF0; s0 < PF0 + AHEAD && s0 < NS; ++s0) request(s0, (s0 - PF0) % (AHEAD + 1));
        if constexpr (PF0 > 0) {
          // the prefetch copies were issued at the first PF_STEPS steps of the GEMM above: everything issued at the
          // later steps (weight copies and jet stores, counted exactly) plus this phase's requests may stay in flight
          constexpr int YOUNGER = chain_ops_in_steps<K1, NS, QD>(PF_STEPS, NTW) + (NS - PF0 < AHEAD ? NS - PF0 : AHEAD) * K1;
          if (ring.issued >= ring.total) wait_vm<0>();      // (end of the slab sequence: copies were skipped)
          else wait_vm<(YOUNGER < 63 ? YOUNGER : 63)>();
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          bf8 cur[K1];
          if (s < PF0) {
#pragma unroll
            for (int c = 0; c < K1; ++c) cur[c] = *reinterpret_cast<const bf8*>(pf + (s * K1 + c) * 1024 + lpos);
          } else {
            if (s + AHEAD < NS) request(s + AHEAD, (s + AHEAD - PF0) % (AHEAD + 1));
#pragma unroll
            for (int c = 0; c < K1; ++c) cur[c] = av[(s - PF0) % (AHEAD + 1)][c];
          }
          const int cs = (s - PF0) % (AHEAD + 1);
          float o[K1][8];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = 4 * h + r;
              const float a = bf2f(cur[0][j]);
              const float sv = fmaf(-a, a, 1.f);
              float cross = 0.f;
              float ab[K1];
#pragma unroll
              for (int c = 0; c < K1; ++c) {
                if constexpr (MEM) ab[c] = bf2f(gv[cs][c][j]);
                else ab[c] = acc[c][2 * s + h][r];
              }
#pragma unroll
              for (int c = 1; c < K1; ++c) {
                cross = fmaf(ab[c], bf2f(cur[c][j]), cross);
                o[c][j] = ab[c] * sv;
              }
              o[0][j] = fmaf(-2.f * a, cross, sv * ab[0]);   // tanh'' = -2 a (1 - a^2)
            }
#pragma unroll
          for (int c = 0; c < K1; ++c) {
            bf8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (__bf16)o[c][j];
            zj[c][s] = v;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (l == nh) phase(std::true_type{});
      else phase(std::false_type{});
      CHAIN_STAMP(3);
      // ---- abar_l = W_l^T zbar_l
      zero_tiles<NTW, K1>(acc);
      const bool pf_cur = live && l >= 2;          // a_l for the next phase (layer l - 1)
      const char* pf_src = reinterpret_cast<const char*>(P.A + (int64_t)(l - 1) * P.jet_stride + tbase);
      bf8 pfh[PFB > 0 ? PFB : 1], pfl[PFB > 0 ? PFB : 1];
      static_for<0, NTW>([&](auto mt_) {
        constexpr int MT = decltype(mt_)::value;
        CHAIN_STAMP(2);
        {
          // zbar_l (= zj, this GEMM's B operand) goes out during this GEMM, two blocks per step, next to one
          // prefetch copy per step.  Steps before R - 1: slab g was requested before the adjoint phase, whose
          // load waits have retired it already (vmcnt retires in order): any count is safe there; later steps
          // count every younger operation exactly.
          constexpr int E = chain_younger_stores<K1, NS, NTW, RB, MT, true, false>();
          constexpr int EP = chain_younger_prefetch<RB, NPF>(MT);
          static_assert((RB - 2) * QD + E + EP <= 63, "vmcnt range");
          if (live && pf_cur) ring.template wait_landed<E + EP, RB>(g, true);
          else if (live) ring.template wait_landed<E, RB>(g, true);
          else ring.template wait_landed<0, RB>(g, false);
        }
        CHAIN_STAMP(0);
        __builtin_amdgcn_s_barrier();
        CHAIN_STAMP(1);
        ring.issue();
        if constexpr (chain_pf_at<NPF>(MT) > 0) {
          if (pf_cur) {                            // copy i = block (piece i / K1, quantity i % K1): piece-major
#pragma unroll
            for (int i = 2 * MT; i < 2 * MT + chain_pf_at<NPF>(MT); ++i)
              dma_1k<PINN_CHAIN_JET_LD_AUX>(pf_src + ((i % K1) * NS + i / K1) * 1024, pf + i * 1024, lane);
          }
        }
        if (live) {
#pragma unroll
          for (int i = 0; i < chain_stores_at<K1, NS>(MT); ++i) {
            const int idx = 2 * MT + i;
            st_blk(zdst, lpos, idx * 1024, zj[idx / NS][idx % NS]);
          }
        }
        const char* sl = ring.consume_ptr();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          constexpr bool havepf = PFB > 0 && MT > 0;
          const bf8 ahi = (havepf && s < PFB) ? pfh[s < PFB ? s : 0] : *reinterpret_cast<const bf8*>(sl + s * 1024);
#pragma unroll
          for (int c = 0; c < K1; ++c) acc[c][MT] = mfma32(ahi, zj[c][s], acc[c][MT]);
          if constexpr (LO) {
            const bf8 alo = (havepf && s < PFB) ? pfl[s < PFB ? s : 0] : *reinterpret_cast<const bf8*>(sl + (NS + s) * 1024);
#pragma unroll
            for (int c = 0; c < K1; ++c) acc[c][MT] = mfma32(alo, zj[c][s], acc[c][MT]);
          }
        }
        ring.consumed();
        if constexpr (PFB > 0 && MT + 1 < NTW) {
          const char* sn = ring.consume_ptr();
#pragma unroll
          for (int s = 0; s < PFB; ++s) {
            pfh[s] = *reinterpret_cast<const bf8*>(sn + s * 1024);
            if constexpr (LO) pfl[s] = *reinterpret_cast<const bf8*>(sn + (NS + s) * 1024);
          }
        }
        ++g;
      });
    }
    // abar_1 for the first layer's reverse kernel (bf16: it is consumed together with the bf16 a_1)
    if (live) {
      const __amdgpu_buffer_rsrc_t g1r = jet_rsrc(P.G1 + tbase, TILE_BYTES);
#pragma unroll
      for (int c = 0; c < K1; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          bf8 v;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * h + r] = (__bf16)acc[c][2 * s + h][r];
          st_blk(g1r, lpos, (c * NS + s) * 1024, v);
        }
    }
    CHAIN_STAMP(4);
  }
  wait_vm<0>();
  CHAIN_DIAG_END(P);
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient of all hidden matrices in one launch.  Workgroup b -> (layer li = b % nh, slice b / nh); wave w
// owns output-unit tiles [w * MTB, (w + 1) * MTB) x all NTW input tiles of dW_l in registers.  The contraction
// runs over (quantity c, point p): per tile two k-steps of 32 — k-step u holds quantities 2u, 2u + 1 (lane group
// qk: quantity 2u + (qk >> 1), points 8 (qk & 1) .. + 7).  Ring unit = (tile, k-step): the zbar and a blocks of two
// quantities, 2 * 2 * NS KB, copied by LDS-DMA; operands by ds_read_b64_tr_b16 (4 points x 16 units per 16 lanes).
__device__ __forceinline__ sh4 ds_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf8 tr_operand(const char* p0) {   // two transposed reads 4 points apart -> 8 k values
  const sh4 a = ds_tr(p0), b = ds_tr(p0 + 4 * 64);
  typedef short sh8 __attribute__((ext_vector_type(8)));
  const sh8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf8, v);
}

// Eight waves, two per SIMD with the step roles of k_chain_fwd8: the kernel streams 2 x 2 GB of jets per layer, and
// with four waves its unit time was set by the 8 copies per wave that open a unit — a wave alone on its SIMD issues
// nothing else while a copy is being accepted — not by its 64 MFMAs (10.5 ms at 12 x 256 / 2^20 points; 8.5 ms so).
template <int NTW, int K1>
__global__ __launch_bounds__(P8_THREADS, 2) void k_chain_wgrad8(const ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = NTW / 2;
  // a wave owns MTB row tiles x NCB column tiles of dW_l (row group wave >> 1, column half wave & 1): 4 x 8 at width 256
  // (12 operands per 32 MFMAs; as 2 x 16 it was 18, and the transposed reads are a measurable part of the unit)
  constexpr int MTB = NTW / 4, NCB = NTW / 2;
  constexpr int KU = (K1 + 1) / 2;                 // k-steps per tile
  constexpr int HALF = 2 * NS * 1024;              // bytes of one operand's two quantities
  constexpr int UNIT = 2 * HALF;                   // [zbar c0 | zbar c1 | a c0 | a c1] x NS blocks
  constexpr int UDMA = UNIT / 1024 / P8_WAVES;     // 1 KB copies per wave and unit
  static_assert(UDMA * P8_WAVES * 1024 == UNIT && MTB >= 1, "unit split over 8 waves");
  constexpr int RU = WG_UNITS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool early = wave < 4;                     // SIMD partners (w, w + 4) order a unit differently: see k_chain_fwd8
  const int i16 = lane & 15, qk = lane >> 4;
  const int nh = P.L - 1;
  const int li = blockIdx.x % nh, slice = blockIdx.x / nh;
  if (slice >= P.n_slices) return;                 // (workgroup-uniform)
  const int64_t t0 = P.n_tiles * slice / P.n_slices, t1 = P.n_tiles * (slice + 1) / P.n_slices;
  const int64_t U = (t1 - t0) * KU;                // ring units this workgroup consumes
  const unsigned short* Zl = P.Z + (int64_t)li * P.jet_stride;        // zbar_{li+1}
  const unsigned short* Al = P.A + (int64_t)li * P.jet_stride;        // a_{li+1} (the layer's input)
  auto issue_unit = [&](int64_t u) {
    if (u >= U) return;
    const int64_t t = t0 + u / KU;
    const int ku = (int)(u % KU);
    char* dst = smem + (int)(u % RU) * UNIT;
    // copy j of this wave: j' = wave * UDMA + j in [0, 4 NS): operand (j' / (2 NS)), quantity 2 ku + (j' / NS) % 2, block j' % NS
#pragma unroll
    for (int j = 0; j < UDMA; ++j) {
      const int jj = wave * UDMA + j;
      const int op = jj / (2 * NS), cq = (jj / NS) & 1, s = jj % NS;
      int c = 2 * ku + cq;
      if (c >= K1) c = K1 - 1;                     // odd K1: the missing quantity is masked at the MFMA operand
      const unsigned short* src = (op ? Al : Zl) + ((t * K1 + c) * NS + s) * 512;
      dma_1k<PINN_CHAIN_JET_LD_AUX>(src, dst + jj * 1024, lane);
    }
  };
  f4 dw[MTB][NCB];
  float bs[MTB];
#pragma unroll
  for (int m = 0; m < MTB; ++m) {
    bs[m] = 0.f;
#pragma unroll
    for (int n = 0; n < NCB; ++n) dw[m][n] = f4{0.f, 0.f, 0.f, 0.f};
  }
  for (int u0 = 0; u0 < RU - 1; ++u0) issue_unit(u0);
  // address of this lane's transposed-read element inside a quantity's NS-block region:
  // rows = points 8 (qk & 1) + (i16 >> 2) (+4 for the second read), column quad = i16 & 3, at fixed (s, h)
  const int tr_lane = (8 * (qk & 1) + (i16 >> 2)) * 64 + (i16 & 3) * 16;
  const int cq_lane = qk >> 1;                     // which of the unit's two quantities this lane group contracts
  CHAIN_DIAG_BEGIN;
  for (int64_t u = 0; u < U; ++u) {
    CHAIN_STAMP(2);
    if (u + RU - 1 > U) wait_vm<0>();              // fewer than RU - 2 younger units exist: drain
    else wait_vm<(RU - 2) * UDMA>();
    CHAIN_STAMP(0);
    __builtin_amdgcn_s_barrier();
    CHAIN_STAMP(1);
    if (early) issue_unit(u + RU - 1);
    const char* ub = smem + (int)(u % RU) * UNIT;
    const int ku = (int)(u % KU);
    const bool qlive = 2 * ku + cq_lane < K1;      // odd K1: the padded quantity contributes nothing
    const char* zb = ub + cq_lane * (NS * 1024) + tr_lane;
    const char* ab = ub + HALF + cq_lane * (NS * 1024) + tr_lane;
    bf8 za[MTB];
#pragma unroll
    for (int m = 0; m < MTB; ++m) {
      const int MT = (wave >> 1) * MTB + m;
      bf8 v = tr_operand(zb + (MT >> 1) * 1024 + (MT & 1) * 8);
      if (!qlive) v = bf8{0, 0, 0, 0, 0, 0, 0, 0};
      za[m] = v;
      if (ku == 0 && qk < 2 && (wave & 1) == 0) {                     // bias gradient: sum over points of zbar's value quantity
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sacc += bf2f(v[j]);
        bs[m] += sacc;
      }
    }
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
      const int NT = (wave & 1) * NCB + n;
      const bf8 bb = tr_operand(ab + (NT >> 1) * 1024 + (NT & 1) * 8);
#pragma unroll
      for (int m = 0; m < MTB; ++m) dw[m][n] = mfma32(za[m], bb, dw[m][n]);
    }
    if (!early) issue_unit(u + RU - 1);
  }
  wait_vm<0>();
  CHAIN_STAMP(2);
  CHAIN_DIAG_END(P);
  // one flush per wave into the flat torch-layout gradient: dW_l (out, in) row-major, then b_l
  float* dWl = P.dW + P.w_off1 + (int64_t)li * P.w_per;
  float* dbl = dWl + (int64_t)P.W * P.W;
#pragma unroll
  for (int m = 0; m < MTB; ++m) {
    const int MT = (wave >> 1) * MTB + m;
#pragma unroll
    for (int n = 0; n < NCB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * MT + 4 * qk + r, col = 16 * ((wave & 1) * NCB + n) + i16;
        if (row < P.W && col < P.W)
          __hip_atomic_fetch_add(dWl + (int64_t)row * P.W + col, dw[m][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    float tsum = bs[m];                            // lanes (i16, qk = 0, 1) hold points 0-7 / 8-15 of unit 16 MT + i16
    tsum += __shfl_xor(tsum, 16, 64);
    const int row = 16 * MT + i16;
    if (qk == 0 && (wave & 1) == 0 && row < P.W) __hip_atomic_fetch_add(dbl + row, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------------------
// First layer, reverse (d_in <= 3): zbar_0 = adjoint(abar_1, a_1) and the layer's gradient
//   dW_0[u][j] = sum_p ( zbar_0(u,p) x_j(p) + sum_c zbar_c(u,p) [dir_col(c) == j] ),   db_0[u] = sum_p zbar_0(u,p)
// in ONE streaming pass over abar_1 (k_chain_bwd's output) and a_1 — 2 x 2 KB per point read, nothing written but
// W x 4 sums.  (The wide engine's two kernels for this wrote zbar_0 back over abar_1 and read it again.)
// A wave walks tiles; lane (p, q) holds the 8 units of k-step s of its point; every product is summed over the tile's
// 16 points with four DPP row rotates, lane p keeps the two sums it owns and adds them to the workgroup's [W][4] table
// in LDS; one global atomic per table entry and workgroup at the end.
__device__ __forceinline__ float dpp_row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

template <int NTW, int K1>
__global__ __launch_bounds__(CHAIN_THREADS, 2) void k_chain_first_bwd(const ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = NTW / 2;
  constexpr int TILE_BYTES = K1 * NS * 1024;
  float* tab = reinterpret_cast<float*>(smem);                 // [16 NTW units][4]: dW_0 columns 0..2, db_0
  for (int i = threadIdx.x; i < 64 * NTW; i += CHAIN_THREADS) tab[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const unsigned lpos = (4u * p + q) * 16u;
  // (the resource covers THIS CHUNK's rows of X: buffer offsets are 32 bits, the whole point set may pass 2 GB)
  const int64_t x_rest = (P.n_points - P.tile0 * 16) * P.d_in * 4;
  const __amdgpu_buffer_rsrc_t xrs = jet_rsrc(P.X + P.tile0 * 16 * P.d_in, (int)(x_rest < 0x7fffffff ? (x_rest > 0 ? x_rest : 0) : 0x7fffffff));
  const int cA = P.dir_col[0], cB = P.dir_col[1], cC = P.dir_col[2];
  for (int64_t t = (int64_t)blockIdx.x * CHAIN_WAVES + wave; t < P.n_tiles; t += (int64_t)gridDim.x * CHAIN_WAVES) {
    const int64_t tbase = uniform64(t) * (K1 * NS * 512);
    const __amdgpu_buffer_rsrc_t gr = jet_rsrc(P.G1 + tbase, TILE_BYTES), ar = jet_rsrc(P.A + tbase, TILE_BYTES);
    const int64_t pt = (P.tile0 + t) * 16 + p;
    float x[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned off = pt < P.n_points && j < P.d_in ? (unsigned)((t * 16 + p) * P.d_in + j) * 4u : 0x7ffffff0u;
      x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)off, 0, 0));
    }
    const bool valid = pt < P.n_points;                        // padding points of the last tile carry no adjoint
#pragma unroll 2
    for (int s = 0; s < NS; ++s) {
      bf8 gv[K1], av[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) { gv[c] = ld_blk(gr, lpos, (c * NS + s) * 1024); av[c] = ld_blk(ar, lpos, (c * NS + s) * 1024); }
      float own0 = 0.f, own1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = bf2f(av[0][j]);
        const float sv = fmaf(-a, a, 1.f);
        float zc[K1], cross = 0.f;
#pragma unroll
        for (int c = 1; c < K1; ++c) { const float g = bf2f(gv[c][j]); cross = fmaf(g, bf2f(av[c][j]), cross); zc[c] = g * sv; }
        zc[0] = fmaf(-2.f * a, cross, sv * bf2f(gv[0][j]));
        float w[4] = {zc[0] * x[0], zc[0] * x[1], zc[0] * x[2], zc[0]};
        if constexpr (K1 > 1) {
          const int dc[3] = {cA, cB, cC};
#pragma unroll
          for (int c = 1; c < K1; ++c) {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) w[jj] += dc[c - 1] == jj ? zc[c] : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float tot = dpp_row_sum16(valid ? w[i] : 0.f);
          const int vidx = j * 4 + i;                          // lane p owns value vidx with (vidx & 15) == p
          if ((vidx >> 4) == 0) own0 = p == (vidx & 15) ? tot : own0;
          else own1 = p == (vidx & 15) ? tot : own1;
        }
      }
      // slot k: value 16 k + p  <->  j = 4 k + (p >> 2), component p & 3  <->  unit 32 s + 16 k + 4 q + (p >> 2)
      atomicAdd(&tab[(32 * s + 4 * q + (p >> 2)) * 4 + (p & 3)], own0);
      atomicAdd(&tab[(32 * s + 16 + 4 * q + (p >> 2)) * 4 + (p & 3)], own1);
    }
  }
  __syncthreads();
  // flat torch layout: W_0 (out, in) row-major at offset 0, then b_0
  for (int i = threadIdx.x; i < 64 * NTW; i += CHAIN_THREADS) {
    const int u = i >> 2, comp = i & 3;
    if (u >= P.W) continue;
    const float v = tab[i];
    if (comp == 3) __hip_atomic_fetch_add(P.dW + (int64_t)P.W * P.d_in + u, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (comp < P.d_in) __hip_atomic_fetch_add(P.dW + (int64_t)u * P.d_in + comp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Last layer, reverse (d_out <= 4): abar_L = W_L^T G (the reverse chain's input, bf16 chain layout) and the layer's gradient
//   dW_L[o][u] = sum over points and quantities G_c[o](p) a_L,c[u](p),   db_L[o] = sum_p G_0[o](p)
// in ONE streaming pass over a_L (2 KB per point read, 2 KB written) — the wide engine's two kernels for this read a_L
// and G separately.  Same shape as k_chain_first_bwd: row sums by DPP, the workgroup's [W][4] table in LDS.
template <int NTW, int K1>
__global__ __launch_bounds__(CHAIN_THREADS, 2) void k_chain_last_bwd(const ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = NTW / 2;
  constexpr int TILE_BYTES = K1 * NS * 1024;
  f4* wlt = reinterpret_cast<f4*>(smem);                        // [16 NTW units]: W_L[0..3][u]
  float* tab = reinterpret_cast<float*>(wlt + 16 * NTW);        // [16 NTW units][4]: dW_L; then db_L[4]
  for (int i = threadIdx.x; i < 16 * NTW; i += CHAIN_THREADS) wlt[i] = *reinterpret_cast<const f4*>(P.WLT + i * 16);
  for (int i = threadIdx.x; i < 64 * NTW + 4; i += CHAIN_THREADS) tab[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const unsigned lpos = (4u * p + q) * 16u;
  for (int64_t t = (int64_t)blockIdx.x * CHAIN_WAVES + wave; t < P.n_tiles; t += (int64_t)gridDim.x * CHAIN_WAVES) {
    const int64_t tbase = uniform64(t) * (K1 * NS * 512);
    const __amdgpu_buffer_rsrc_t ar = jet_rsrc(P.A + tbase, TILE_BYTES), glr = jet_rsrc(P.GL + tbase, TILE_BYTES);
    f4 G[K1];
#pragma unroll
    for (int c = 0; c < K1; ++c) G[c] = *reinterpret_cast<const f4*>(P.gout + ((uniform64(t) * K1 + c) * 256 + p * 4));
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float tot = dpp_row_sum16(G[0][o]);
      if (lane == 0) atomicAdd(&tab[64 * NTW + o], tot);
    }
#pragma unroll 2
    for (int s = 0; s < NS; ++s) {
      bf8 av[K1], gb[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) av[c] = ld_blk(ar, lpos, (c * NS + s) * 1024);
      float own0 = 0.f, own1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f4 wl = wlt[32 * s + 16 * (j >> 2) + 4 * q + (j & 3)];
#pragma unroll
        for (int c = 0; c < K1; ++c)
          gb[c][j] = (__bf16)fmaf(wl[3], G[c][3], fmaf(wl[2], G[c][2], fmaf(wl[1], G[c][1], wl[0] * G[c][0])));
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float w = 0.f;
#pragma unroll
          for (int c = 0; c < K1; ++c) w = fmaf(G[c][o], bf2f(av[c][j]), w);
          const float tot = dpp_row_sum16(w);
          const int vidx = j * 4 + o;                          // lane p owns value vidx with (vidx & 15) == p
          if ((vidx >> 4) == 0) own0 = p == (vidx & 15) ? tot : own0;
          else own1 = p == (vidx & 15) ? tot : own1;
        }
      }
#pragma unroll
      for (int c = 0; c < K1; ++c) st_blk(glr, lpos, (c * NS + s) * 1024, gb[c]);
      atomicAdd(&tab[(32 * s + 4 * q + (p >> 2)) * 4 + (p & 3)], own0);
      atomicAdd(&tab[(32 * s + 16 + 4 * q + (p >> 2)) * 4 + (p & 3)], own1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * NTW + 4; i += CHAIN_THREADS) {
    const float v = tab[i];
    if (i >= 64 * NTW) { if (i - 64 * NTW < P.d_out) __hip_atomic_fetch_add(P.dWL + (int64_t)P.d_out * P.W + (i - 64 * NTW), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if ((i >> 2) < P.W && (i & 3) < P.d_out)
      __hip_atomic_fetch_add(P.dWL + (int64_t)(i & 3) * P.W + (i >> 2), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Output layer, forward: out = W_L a_L + b_L on the bf16 a_L (chain layout) with v_mfma_f32_16x16x32_bf16 — W_L split
// hi + lo like the hidden weights — then the loss epilogue of the fused / wide kernels (outputs, residual, fidelity
// terms, loss partial sums, output adjoint G).  The wide engine's kernel for this converted a_L to fp32 and ran 256
// fp32 MFMAs per tile: 8 k cycles of matrix pipe and ~3 k of conversions that the fp32 MFMA cannot overlap, next to an
// epilogue of ~10 k (0.87 ms per 2^20 points); here the product is 64 bf16 MFMAs on operands that need no conversion.
constexpr int LAST_PADS = 4;        // LDS pads per wave for the epilogue's scatter (it uses at most two)
template <int NTW, int K1, bool GRAD>
__global__ __launch_bounds__(WIDE_THREADS, 2) void k_chain_last_fwd(const FusedParams P, const WideLayer Lp) {
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  float* smem = reinterpret_cast<float*>(smem_);
  constexpr int NS = NTW / 2;
  constexpr int TILE_BYTES = K1 * NS * 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  bf8* wf = reinterpret_cast<bf8*>(smem);                                  // [hi | lo][k-step s][lane]: A fragments of W_L
  float* tb = smem + 2 * NS * 64 * 4 + wave * (LAST_PADS * TB_FLOATS);
  float* lsum = smem + 2 * NS * 64 * 4 + WIDE_WAVES * LAST_PADS * TB_FLOATS;
  for (int i = threadIdx.x; i < NS * 64; i += WIDE_THREADS) {
    const int s = i >> 6, ln = i & 63, o = ln & 15, qk = ln >> 4;
    bf8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = Lp.W[o * (16 * NTW) + 32 * s + 16 * (j >> 2) + 4 * qk + (j & 3)];   // padded row-major [16][WP]
      hi[j] = (__bf16)w;
      lo[j] = (__bf16)(w - (float)hi[j]);
    }
    wf[i] = hi; wf[NS * 64 + i] = lo;
  }
  __syncthreads();
  float sums[MAX_SUMS];
#pragma unroll
  for (int j = 0; j < MAX_SUMS; ++j) sums[j] = 0.f;
  ScatterMap<K1> sm, sm_mse;
  build_scatter_maps<K1>(P, q, sm, sm_mse);
  const unsigned lpos = (4u * p + q) * 16u;
  const unsigned short* AL = reinterpret_cast<const unsigned short*>(Lp.in_act);
  const int gw = blockIdx.x * WIDE_WAVES + wave, nw = gridDim.x * WIDE_WAVES;
  for (int64_t t = gw; t < Lp.n_tiles; t += nw) {
    const int64_t pt = (Lp.tile0 + t) * 16 + p;
    const bool valid = pt < P.N;
    const int64_t ptc = valid ? pt : P.N - 1;
    const __amdgpu_buffer_rsrc_t ar = jet_rsrc(AL + uniform64(t) * (K1 * NS * 512), TILE_BYTES);
    f4 acc[K1][1];
    init_bias<1, K1>(Lp.b, acc, q);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bf8 whi = wf[s * 64 + lane], wlo = wf[(NS + s) * 64 + lane];
#pragma unroll
      for (int c = 0; c < K1; ++c) {
        const bf8 b = ld_blk(ar, lpos, (c * NS + s) * 1024);
        acc[c][0] = mfma32(whi, b, acc[c][0]);
        acc[c][0] = mfma32(wlo, b, acc[c][0]);
      }
    }
    f4 G[K1][1];
    loss_epilogue<K1, GRAD>(P, acc, G, sums, sm, sm_mse, tb, pt, ptc, valid, p, q);
    if constexpr (GRAD) {
#pragma unroll
      for (int c = 0; c < K1; ++c) *reinterpret_cast<f4*>(Lp.g_out + ((t * K1 + c) * 1 + 0) * 256 + lane * 4) = G[c][0];
    }
  }
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

template <int NTW> int launch_chain_fwd8(int K1, bool fold_first, const ChainParams& P, int grid, hipStream_t s);
template <int NTW> int launch_chain_bwd(int K1, const ChainParams& P, int grid, hipStream_t s);
template <int NTW> int launch_chain_wgrad8(int K1, const ChainParams& P, int grid, hipStream_t s);
template <int NTW> int launch_chain_first_bwd(int K1, const ChainParams& P, int grid, hipStream_t s);
template <int NTW> int launch_chain_last_bwd(int K1, const ChainParams& P, int grid, hipStream_t s);
template <int NTW> int launch_chain_last_fwd(int K1, bool grad, const FusedParams& P, const WideLayer& Lp, int grid, hipStream_t s);

}  // namespace pinn
