// pinn_wide.hip — host side of the wide MFMA engine (64 < W <= 256): weight packing, chunked
// point loop, per-layer launches.  Kernels: wide_kernel.h.
#include <type_traits>
#include "chain_kernel.h"

namespace pinn {

namespace {

int wide_padded_width(int W) { return W <= 128 ? 128 : 256; }

struct WGeo { int WP, NTW, PW, PB; };
WGeo wgeo(const Net& n) {
  WGeo g;
  g.WP = wide_padded_width(n.W); g.NTW = g.WP / 16;
  g.PW = g.WP * 16 + (n.L - 1) * g.WP * g.WP + 16 * g.WP;
  g.PB = n.L * g.WP + 16;
  return g;
}
int64_t al256(int64_t v) { return (v + 255) & ~(int64_t)255; }
// bytes between the 1 KB pieces of a packed weight slab (k_chain_pack): all slabs' piece i, rounded up to 64 KB, + 4 KB
int64_t chain_plane_bytes(int nh, int NTW) {
  const int64_t slabs = (int64_t)(nh > 0 ? nh : 1) * NTW * 1024;
  return ((slabs + 65535) / 65536) * 65536 + 4096;
}

int cus() { return device_cu_count(); }

// bf16 mode's output-layer kernel (k_chain_last_fwd: three waves per SIMD fit) runs up to three workgroups per CU, each
// with its own row of loss partial sums
constexpr int SUM_ROWS_PER_CU = 3;
constexpr int64_t ACT_BUDGET_BYTES = (int64_t)6 << 30;   // activation workspace per chunk (fp32 mode)
// bf16 mode (chain kernels): 2L + 1 bf16 jets per chunk, and the weight-gradient kernel flushes its registers
// once per chunk and workgroup — chunks are as large as a 288 GB part comfortably allows
constexpr int64_t CHAIN_BUDGET_BYTES = (int64_t)56 << 30;

struct WLayout {
  int64_t chunk_pts, chunk_tiles, n_chunks;
  int64_t wp, wtp, bp, wp16, wtp16, act, act_stride, gA, gB, gZ, gout, sums, total;
  int64_t wf, wtf, jA, jZ, jGL, jG1, jet_stride;   // bf16 mode: packed fragments, chain-layout jets (byte offsets; jet_stride in bytes)
  int grid;
};
WLayout wlayout(const Net& n, const WGeo& g, int64_t N) {
  WLayout w;
  const int K1 = 1 + PINN_MAX_DIRS;
  const bool chain = n.prec == PINN_PREC_BF16;
  // fp32 mode: L jets + 2 adjoint buffers, fp32.  bf16 mode: a_1..a_L, zbar_1..zbar_{L-1}, abar_L, abar_1 in bf16.
  const int64_t per_pt = chain ? (int64_t)K1 * g.WP * 2 * (2 * n.L + 1) + (int64_t)K1 * 16 * 4
                               : (int64_t)K1 * g.WP * 4 * (n.L + 2) + (int64_t)K1 * 16 * 4;
  // whole number of tiles per wave in every full chunk (no tail imbalance): multiple of waves * 16 points
  const int64_t quantum = (int64_t)cus() * WIDE_WAVES * 16;
  int64_t cp = (chain ? CHAIN_BUDGET_BYTES : ACT_BUDGET_BYTES) / per_pt;
  cp = (cp / quantum) * quantum;
  if (cp < quantum) cp = quantum;
  const int64_t npad = ((N + 15) / 16) * 16;
  if (cp > npad) cp = npad;
  w.chunk_pts = cp; w.chunk_tiles = cp / 16; w.n_chunks = (npad + cp - 1) / cp;
  w.grid = cus();
  int64_t off = 0;
  w.wp = off; off += al256((int64_t)g.PW * 4);
  w.wtp = off; off += al256((int64_t)g.PW * 4);
  w.bp = off; off += al256((int64_t)g.PB * 4);
  w.wp16 = off; off += al256((int64_t)g.PW * 2);
  w.wtp16 = off; off += al256((int64_t)g.PW * 2);
  w.act_stride = al256(w.chunk_tiles * K1 * g.NTW * 256 * 4);
  w.act = w.gA = w.gB = w.gZ = 0;
  w.wf = w.wtf = w.jA = w.jZ = w.jGL = w.jG1 = w.jet_stride = 0;
  if (!chain) {
    w.act = off; off += w.act_stride * n.L;
    w.gA = off; off += w.act_stride;
    w.gB = off; off += w.act_stride;
  } else {
    const int64_t frag = chain_plane_bytes(n.L - 1, g.NTW) * g.NTW;          // hi + lo bf16 per hidden matrix: 2 NS = NTW piece planes
    w.wf = off; off += frag;
    w.wtf = off; off += frag;
    w.jet_stride = al256(w.chunk_tiles * K1 * g.NTW * 256 * 2);
    w.jA = off; off += w.jet_stride * n.L;
    w.jZ = off; off += w.jet_stride * (n.L > 1 ? n.L - 1 : 0);
    w.jGL = off; off += w.jet_stride;
    if (n.L > 1) { w.jG1 = off; off += w.jet_stride; }
    else w.jG1 = w.jGL;                                  // no hidden matrix: abar_1 is abar_L
  }
  w.gout = off; off += al256(w.chunk_tiles * K1 * 256 * 4);
  w.sums = off; off += al256(w.n_chunks * SUM_ROWS_PER_CU * w.grid * MAX_SUMS * 4);   // one row of loss partials per workgroup
  w.total = off;
  return w;
}

__device__ inline unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

__global__ void k_wide_pack(Net n, int WP, const float* __restrict__ params, float* __restrict__ Wp,
                            float* __restrict__ WTp, float* __restrict__ Bp, unsigned short* __restrict__ Wp16,
                            unsigned short* __restrict__ WTp16, int PW, int PB) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < PW) {
    int l, rem;
    if (i < WP * 16) { l = 0; rem = i; }
    else { const int j = i - WP * 16; l = 1 + j / (WP * WP); rem = j % (WP * WP); if (l > n.L) l = n.L; }
    const int inP = (l == 0) ? 16 : WP, outP = (l == n.L) ? 16 : WP;
    const int in_d = n.in_dim(l), out_d = n.out_dim(l);
    const int base = i - rem;
    { const int o = rem / inP, c = rem % inP;
      const float v = (o < out_d && c < in_d) ? params[n.w_off(l) + (int64_t)o * in_d + c] : 0.f;
      Wp[i] = v; Wp16[i] = bf16_bits(v); }
    { const int c = rem / outP, o = rem % outP;
      const float v = (o < out_d && c < in_d) ? params[n.w_off(l) + (int64_t)o * in_d + c] : 0.f;
      WTp[base + rem] = v; WTp16[base + rem] = bf16_bits(v); }
  }
  if (i < PB) {
    int l = i / WP, o = i % WP;
    if (l >= n.L) { l = n.L; o = i - n.L * WP; }
    Bp[i] = (o < n.out_dim(l)) ? params[n.b_off(l) + o] : 0.f;
  }
}

__global__ void k_wide_reduce_sums(const float* __restrict__ wg_sums, int64_t rows, int col0, int nt,
                                   float* __restrict__ out) {
  const int t = col0 + blockIdx.x;
  __shared__ double red[256];
  double v = 0.0;
  for (int64_t b = threadIdx.x; b < rows; b += 256) v += (double)wg_sums[b * MAX_SUMS + t];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0 && blockIdx.x < nt) out[blockIdx.x] = (float)red[0];
}

// Hidden matrices W_1 .. W_{L-1} -> MFMA fragment order for the chain kernels (chain_kernel.h), hi + lo bf16:
//   Wf [li][MT][hl][s][lane][j] = W_l[16 MT + m][n(s, qk, j)]        (forward: rows = output units)
//   WTf[li][MT][hl][s][lane][j] = W_l[n(s, qk, j)][16 MT + m]        (reverse: rows = input units)
// lane = (m = lane & 15, qk = lane >> 4), n(s, qk, j) = 32 s + 16 (j >> 2) + 4 qk + (j & 3): the k index is
// permuted inside each k-step so that two accumulator tiles are the next B operand as they stand.
// A slab (li, MT) is 2 NS pieces of 1 KB ([hi | lo][k-step]); piece i of slab n is stored at element
// i * plane + n * 512: the pieces of one slab are `plane` apart (a multiple of 64 KB plus 4 KB, chain_plane_bytes), so
// that the 32 CUs of an XCD, which read the same slab at about the same time, spread over the L2's channels
// instead of queueing on the few that one contiguous 16 KB region maps to.
__global__ void k_chain_pack(Net n, int NTW, int64_t plane, const float* __restrict__ params, unsigned short* __restrict__ Wf,
                             unsigned short* __restrict__ WTf) {
  const int NS = NTW / 2;
  const int64_t per_layer = (int64_t)NTW * NS * 512;            // (MT, s, lane, j) combinations
  const int64_t total = (int64_t)(n.L - 1) * per_layer;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int li = (int)(e / per_layer);
  int64_t r = e % per_layer;
  const int MT = (int)(r / (NS * 512)); r %= NS * 512;
  const int s = (int)(r / 512); r %= 512;
  const int lane = (int)(r / 8), j = (int)(r % 8);
  const int m = lane & 15, qk = lane >> 4;
  const int row = 16 * MT + m, kk = 32 * s + 16 * (j >> 2) + 4 * qk + (j & 3);
  const int W = n.W;
  const float* Wl = params + n.w_off(li + 1);
  const bool in = row < W && kk < W;
  const float wf = in ? Wl[(int64_t)row * W + kk] : 0.f;
  const float wt = in ? Wl[(int64_t)kk * W + row] : 0.f;
  const int64_t slab = ((int64_t)li * NTW + MT) * 512;
  const int64_t o_hi = (int64_t)s * plane + slab + lane * 8 + j, o_lo = o_hi + (int64_t)NS * plane;
  const __bf16 fh = (__bf16)wf, th = (__bf16)wt;
  Wf[o_hi] = __builtin_bit_cast(unsigned short, fh);
  Wf[o_lo] = __builtin_bit_cast(unsigned short, (__bf16)(wf - (float)fh));
  WTf[o_hi] = __builtin_bit_cast(unsigned short, th);
  WTf[o_lo] = __builtin_bit_cast(unsigned short, (__bf16)(wt - (float)th));
}

template <int NTW>
int run_w(const Net& n, const WGeo& g, bool grad, const LossReq* rq, const float* params, const float* X, int64_t N,
          float* Y, float* dY, char* base, const WLayout& w, hipStream_t s) {
  FusedParams P;
  memset(&P, 0, sizeof(P));
  P.d_in = n.d_in; P.d_out = n.d_out; P.L = n.L; P.act = n.act;
  for (int j = 0; j < PINN_MAX_DIRS; ++j) P.dir_col[j] = n.dir_col[j];
  P.N = N; P.X = X; P.Y = Y; P.dY = dY;
  P.n_split = rq ? rq->n_split : -1;
  P.wg_sums = (float*)(base + w.sums);
  if (rq) {
    P.loss_kind = rq->kind == 0 ? 1 : (rq->kind == 1 ? 2 : 3);
    if (P.loss_kind & 1) {
      P.scale = rq->scale;
      P.residual_id = rq->spec.residual_id;
      for (int j = 0; j < PINN_MAX_ROLES; ++j) P.out_col[j] = rq->spec.out_col[j];
      for (int d = 0; d < PINN_MAX_DIRS; ++d) P.q_of[d] = 1 + rq->spec.dir_of[d];
      P.thr = rq->spec.param[0]; P.anchor = rq->spec.param[1];
      P.xcol = n.dir_col[rq->spec.dir_of[0]];
    } else {
      for (int j = 0; j < PINN_MAX_ROLES; ++j) P.out_col[j] = -1;
    }
    if (P.loss_kind & 2) {
      P.n_cols = rq->n_cols; P.T = rq->T; P.mse_scale = rq->mse_scale;
      for (int j = 0; j < PINN_MAX_ROLES; ++j) P.mse_col[j] = j < rq->n_cols ? rq->out_col[j] : -1;
    }
  }
  const float* Wp = (const float*)(base + w.wp);
  const float* WTp = (const float*)(base + w.wtp);
  const float* Bp = (const float*)(base + w.bp);
  const unsigned short* Wp16 = (const unsigned short*)(base + w.wp16);
  const unsigned short* WTp16 = (const unsigned short*)(base + w.wtp16);
  const int prec = n.prec;
  const int packN = g.PW > g.PB ? g.PW : g.PB;
  hipLaunchKernelGGL(k_wide_pack, dim3((packN + 255) / 256), dim3(256), 0, s, n, g.WP, params, (float*)(base + w.wp),
                     (float*)(base + w.wtp), (float*)(base + w.bp), (unsigned short*)(base + w.wp16),
                     (unsigned short*)(base + w.wtp16), g.PW, g.PB);
  const bool chain = prec == PINN_PREC_BF16;
  const int nh = n.L - 1;                                  // hidden (W x W) matrices
  if (chain && nh > 0) {
    const int64_t total = (int64_t)nh * NTW * (NTW / 2) * 512;
    hipLaunchKernelGGL(k_chain_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n, NTW, chain_plane_bytes(nh, NTW) / 2, params,
                       (unsigned short*)(base + w.wf), (unsigned short*)(base + w.wtf));
  }
  auto woff = [&](int l) { return l == 0 ? 0 : g.WP * 16 + (l - 1) * g.WP * g.WP; };
  auto act_l = [&](int l) { return (float*)(base + w.act + (int64_t)(l - 1) * w.act_stride); };   // a_l, l = 1..L
  float* gA = (float*)(base + w.gA);
  float* gB = (float*)(base + w.gB);
  float* gout = (float*)(base + w.gout);
  const int K1 = n.K1, L = n.L;
  const int64_t total_tiles = (N + 15) / 16;
  int rc = PINN_OK;
  for (int64_t ch = 0; ch < w.n_chunks && rc == PINN_OK; ++ch) {
    WideLayer Lp;
    memset(&Lp, 0, sizeof(Lp));
    Lp.tile0 = ch * w.chunk_tiles;
    Lp.n_tiles = total_tiles - Lp.tile0 < w.chunk_tiles ? total_tiles - Lp.tile0 : w.chunk_tiles;
    if (Lp.n_tiles <= 0) break;
    Lp.sums_slot = (int)(ch * SUM_ROWS_PER_CU * w.grid);
    const int grid = (int)((Lp.n_tiles + WIDE_WAVES - 1) / WIDE_WAVES < w.grid ? (Lp.n_tiles + WIDE_WAVES - 1) / WIDE_WAVES : w.grid);
    // zero this chunk's rows of the partial-sum table (a smaller grid leaves rows untouched)
    if (hipMemsetAsync(P.wg_sums + (int64_t)Lp.sums_slot * MAX_SUMS, 0, (size_t)SUM_ROWS_PER_CU * w.grid * MAX_SUMS * 4, s) != hipSuccess) {
      set_error("hipMemsetAsync failed"); return PINN_ERR_LAUNCH;
    }
    if (chain) {
      // ---- bf16 mode: first / last layer on this file's kernels (fp32 MFMA on the thin matrices, chain-layout
      // bf16 jets), the L - 1 hidden matrices on the three chain kernels (chain_kernel.h) ----
      ChainParams C;
      memset(&C, 0, sizeof(C));
      C.L = L; C.n_tiles = Lp.n_tiles; C.jet_stride = w.jet_stride / 2; C.w_plane = chain_plane_bytes(nh, NTW);
      C.Wf = (const unsigned short*)(base + w.wf); C.WTf = (const unsigned short*)(base + w.wtf); C.bias = Bp;
      C.A = (unsigned short*)(base + w.jA); C.Z = (unsigned short*)(base + w.jZ);
      C.GL = (unsigned short*)(base + w.jGL); C.G1 = (unsigned short*)(base + w.jG1);
      C.dW = grad ? rq->grad : nullptr; C.spill = grad ? 1 : 0; C.W = n.W;
      C.w_off1 = n.w_off(1); C.w_per = (int64_t)n.W * n.W + n.W;
#ifdef PINN_CHAIN_DIAG
      static unsigned long long* dbuf = nullptr;     // diagnostic build only (never shipped): 3 kernels x 8 phase sums
      if (!dbuf) { (void)hipMalloc((void**)&dbuf, 48 * sizeof(unsigned long long)); }
#endif
      auto jetA = [&](int l) { return (float*)(base + w.jA + (int64_t)(l - 1) * w.jet_stride); };   // a_l, l = 1..L
      const int cgrid = (int)((Lp.n_tiles + CHAIN_WAVES - 1) / CHAIN_WAVES < w.grid ? (Lp.n_tiles + CHAIN_WAVES - 1) / CHAIN_WAVES : w.grid);
      // d_in <= 3 (every network of the reference): the first layer is folded into the forward chain kernel
      const bool fold_first = nh > 0 && n.d_in <= 3;
      C.X = X; C.W0 = Wp; C.n_points = N; C.tile0 = Lp.tile0; C.d_in = n.d_in;
      for (int j = 0; j < 3; ++j) C.dir_col[j] = j < n.k ? n.dir_col[j] : 0;
      Lp.W = Wp; Lp.b = Bp; Lp.out_act = jetA(1);
      if (!fold_first) { rc = launch_wide_fwd<NTW>(0, K1, prec, false, P, Lp, grid, s); if (rc) break; }
#ifdef PINN_CHAIN_DIAG
      C.diag = dbuf;
#endif
      if (nh > 0) { rc = launch_chain_fwd8<NTW>(K1, fold_first, C, cgrid, s); if (rc) break; }
      Lp.W = Wp + woff(L); Lp.b = Bp + L * g.WP; Lp.in_act = jetA(L); Lp.out_act = nullptr; Lp.g_out = gout;
      {
        const int64_t want = (Lp.n_tiles + WIDE_WAVES - 1) / WIDE_WAVES;
        const int g3 = (int)(want < (int64_t)SUM_ROWS_PER_CU * w.grid ? want : (int64_t)SUM_ROWS_PER_CU * w.grid);
        rc = launch_chain_last_fwd<NTW>(K1, grad, P, Lp, g3, s); if (rc) break;
      }
#ifdef PINN_CHAIN_DIAG
      if (!grad && nh > 0) {
        unsigned long long h[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[i];
        fprintf(stderr, "CHAIN_DIAG fwd (no spill): wait_dma %.1f%% barrier %.1f%% step %.1f%% act %.1f%% tile-io %.1f%% | total %llu cycles\n",
                100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot, tot);
      }
#endif
      if (!grad) continue;
      // output layer: dW_L = G . a_L^T ; abar_L = W_L^T G (bf16, chain layout) — one streaming kernel when d_out <= 4
      if (n.d_out <= 4) {
        ChainParams F = C;
        F.A = (unsigned short*)jetA(L); F.gout = gout; F.WLT = WTp + woff(L); F.dWL = rq->grad + n.w_off(L); F.d_out = n.d_out;
        const int64_t want = (Lp.n_tiles + CHAIN_WAVES - 1) / CHAIN_WAVES;
        rc = launch_chain_last_bwd<NTW>(K1, F, (int)(want < 8 * (int64_t)w.grid ? want : 8 * (int64_t)w.grid), s); if (rc) break;
      } else {
        Lp.g_in = gout; Lp.in_act = jetA(L); Lp.in_d = n.in_dim(L); Lp.out_d = n.out_dim(L);
        Lp.dW = rq->grad + n.w_off(L); Lp.db = rq->grad + n.b_off(L);
        rc = launch_wide_wgrad<NTW>(2, K1, prec, P, Lp, w.grid, s); if (rc) break;
        Lp.W = WTp + woff(L); Lp.g_out = (float*)C.GL;
        rc = launch_wide_bwd<NTW>(2, K1, prec, P, Lp, grid, s); if (rc) break;
      }
#ifdef PINN_CHAIN_DIAG
      C.diag = dbuf + 8;
#endif
      if (nh > 0) { rc = launch_chain_bwd<NTW>(K1, C, cgrid, s); if (rc) break; }
      if (fold_first) {
        // first layer: zbar_0 = adjoint(abar_1, a_1) and dW_0, db_0 in one streaming pass (k_chain_first_bwd)
        ChainParams F = C;
        F.A = (unsigned short*)jetA(1); F.dW = rq->grad;
        const int64_t want = (Lp.n_tiles + CHAIN_WAVES - 1) / CHAIN_WAVES;
        rc = launch_chain_first_bwd<NTW>(K1, F, (int)(want < 8 * (int64_t)w.grid ? want : 8 * (int64_t)w.grid), s); if (rc) break;
      } else {
        // first layer: zbar_0 = adjoint(abar_1, a_1), in place over abar_1; dW_0 = zbar_0 . (x, e_j)^T
        Lp.g_in = (float*)C.G1; Lp.in_act = jetA(1); Lp.g_out = nullptr; Lp.W = nullptr; Lp.z_out = (float*)C.G1;
        rc = launch_wide_bwd<NTW>(0, K1, prec, P, Lp, grid, s); if (rc) break;
        Lp.g_in = (float*)C.G1; Lp.in_act = nullptr; Lp.in_d = n.in_dim(0); Lp.out_d = n.out_dim(0);
        Lp.dW = rq->grad + n.w_off(0); Lp.db = rq->grad + n.b_off(0);
        rc = launch_wide_wgrad<NTW>(0, K1, prec, P, Lp, w.grid, s); if (rc) break;
      }
      if (nh > 0) {
        C.n_slices = w.grid / nh > 0 ? w.grid / nh : 1;
        if ((int64_t)C.n_slices > Lp.n_tiles) C.n_slices = (int)Lp.n_tiles;
#ifdef PINN_CHAIN_DIAG
        C.diag = dbuf + 16;
#endif
        rc = launch_chain_wgrad8<NTW>(K1, C, C.n_slices * nh, s); if (rc) break;
#ifdef PINN_CHAIN_DIAG
        {
          unsigned long long h[48];
          (void)hipStreamSynchronize(s);
          (void)hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
          for (int k8 = 0; k8 < 2; ++k8) {
            const unsigned long long* d = h + (k8 == 0 ? 0 : 24);
            unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += d[i];
            fprintf(stderr, "CHAIN_DIAG %s %s wave: wait %.1f%% barrier %.1f%% reads+mfma %.1f%% act %.1f%% copies+stores %.1f%% other %.1f%% tile-io %.1f%% | total %llu cycles\n",
                    k8 < 2 ? "fwd8" : "bwd8", (k8 & 1) ? "late" : "early", 100.0 * d[0] / tot, 100.0 * d[1] / tot, 100.0 * d[2] / tot, 100.0 * d[3] / tot, 100.0 * d[4] / tot, 100.0 * d[5] / tot, 100.0 * d[6] / tot, tot);
          }
          const char* nm[3] = {"fwd", "bwd", "wgrad"};
          for (int k = 0; k < 3; ++k) {
            unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[8 * k + i];
            fprintf(stderr, "CHAIN_DIAG %s: wait_dma %.1f%% barrier %.1f%% step(issue+lds+mfma) %.1f%% act/adjoint %.1f%% tile-io %.1f%% | total %llu cycles\n", nm[k],
                    100.0 * h[8 * k] / tot, 100.0 * h[8 * k + 1] / tot, 100.0 * h[8 * k + 2] / tot, 100.0 * h[8 * k + 3] / tot, 100.0 * h[8 * k + 4] / tot, tot);
          }
        }
#endif
      }
      continue;
    }
    // ---- forward ----
    Lp.W = Wp; Lp.W16 = Wp16; Lp.b = Bp; Lp.out_act = act_l(1);
    rc = launch_wide_fwd<NTW>(0, K1, prec, false, P, Lp, grid, s); if (rc) break;
    for (int l = 1; l < L; ++l) {
      Lp.W = Wp + woff(l); Lp.W16 = Wp16 + woff(l); Lp.b = Bp + l * g.WP; Lp.in_act = act_l(l); Lp.out_act = act_l(l + 1);
      rc = launch_wide_fwd<NTW>(1, K1, prec, false, P, Lp, grid, s); if (rc) break;
    }
    if (rc) break;
    Lp.W = Wp + woff(L); Lp.W16 = Wp16 + woff(L); Lp.b = Bp + L * g.WP; Lp.in_act = act_l(L); Lp.out_act = nullptr; Lp.g_out = gout;
    rc = launch_wide_fwd<NTW>(2, K1, prec, grad, P, Lp, grid, s); if (rc) break;
    if (!grad) continue;
    // ---- reverse sweep ----
    const int gx_h = w.grid;   // row blocks are waves of a workgroup now; one workgroup per CU
    // output layer: dW_L = G . a_L^T ; abar_L = W_L^T G
    Lp.g_in = gout; Lp.in_act = act_l(L); Lp.in_d = n.in_dim(L); Lp.out_d = n.out_dim(L);
    Lp.dW = rq->grad + n.w_off(L); Lp.db = rq->grad + n.b_off(L);
    rc = launch_wide_wgrad<NTW>(2, K1, prec, P, Lp, w.grid, s); if (rc) break;
    Lp.W = WTp + woff(L); Lp.W16 = WTp16 + woff(L); Lp.g_out = gA;
    rc = launch_wide_bwd<NTW>(2, K1, prec, P, Lp, grid, s); if (rc) break;
    float* gcur = gA; float* gnext = gB;
    for (int l = L - 1; l >= 1; --l) {
      // zbar_l (in place over gcur) and abar_l = W_l^T zbar_l
      Lp.W = WTp + woff(l); Lp.W16 = WTp16 + woff(l); Lp.g_in = gcur; Lp.in_act = act_l(l + 1); Lp.g_out = gnext;
      float* zdst = gcur;   // one wave per tile -> zbar goes back in place over the incoming adjoint
      Lp.z_out = zdst;
      rc = launch_wide_bwd<NTW>(1, K1, prec, P, Lp, grid, s); if (rc) break;
      Lp.g_in = zdst; Lp.in_act = act_l(l); Lp.in_d = n.in_dim(l); Lp.out_d = n.out_dim(l);
      Lp.dW = rq->grad + n.w_off(l); Lp.db = rq->grad + n.b_off(l);
      rc = launch_wide_wgrad<NTW>(1, K1, prec, P, Lp, gx_h, s); if (rc) break;
      float* t = gcur; gcur = gnext; gnext = t;
    }
    if (rc) break;
    Lp.g_in = gcur; Lp.in_act = act_l(1); Lp.g_out = nullptr; Lp.W = nullptr; Lp.z_out = gcur;
    rc = launch_wide_bwd<NTW>(0, K1, prec, P, Lp, grid, s); if (rc) break;
    Lp.g_in = gcur; Lp.in_act = nullptr; Lp.in_d = n.in_dim(0); Lp.out_d = n.out_dim(0);
    Lp.dW = rq->grad + n.w_off(0); Lp.db = rq->grad + n.b_off(0);
    rc = launch_wide_wgrad<NTW>(0, K1, prec, P, Lp, gx_h, s); if (rc) break;
  }
  if (rc) return rc;
  if (rq) {
    if (P.loss_kind & 1)
      hipLaunchKernelGGL(k_wide_reduce_sums, dim3(rq->n_terms), dim3(256), 0, s, (const float*)P.wg_sums,
                         (int64_t)w.n_chunks * SUM_ROWS_PER_CU * w.grid, 0, rq->n_terms, rq->sums);
    if (P.loss_kind & 2)
      hipLaunchKernelGGL(k_wide_reduce_sums, dim3(rq->n_cols), dim3(256), 0, s, (const float*)P.wg_sums,
                         (int64_t)w.n_chunks * SUM_ROWS_PER_CU * w.grid, MSE_SUM0, rq->n_cols, rq->mse_sums);
  }
  return check_launch("wide reductions");
}

int run(const Net& n, bool grad, const LossReq* rq, const float* params, const float* X, int64_t N, float* Y,
        float* dY, void* ws, int64_t ws_bytes, hipStream_t s) {
  const WGeo g = wgeo(n);
  const WLayout w = wlayout(n, g, N);
  if (!ws || ws_bytes < w.total) {
    set_error("workspace too small: need %lld bytes, got %lld", (long long)w.total, (long long)ws_bytes);
    return PINN_ERR_WORKSPACE;
  }
  return g.NTW == 8 ? run_w<8>(n, g, grad, rq, params, X, N, Y, dY, (char*)ws, w, s)
                    : run_w<16>(n, g, grad, rq, params, X, N, Y, dY, (char*)ws, w, s);
}

}  // namespace

bool wide_supports(const Net& n) {
  // bf16 mode: k_chain_fwd8 keeps every layer's bias in LDS behind its weight ring; the depth limit is its LDS formula
  if (n.prec == PINN_PREC_BF16 && chain_fwd8_lds_bytes(n.W <= 128 ? 8 : 16, n.L, true) > CHAIN_LDS_LIMIT) return false;
  return n.act == PINN_ACT_TANH && n.W > 64 && n.W <= 256 && n.d_in <= 16 && n.d_out <= 16 && n.L >= 1 &&
         (n.K1 == 1 || n.K1 == 3 || n.K1 == 4);
}

int64_t wide_workspace_bytes(const Net& n, int64_t N) {
  Net m = n;   // one workspace serves the k = 0 calls too
  if (!(n.act == PINN_ACT_TANH && n.W > 64 && n.W <= 256 && n.d_in <= 16 && n.d_out <= 16)) return -1;
  return wlayout(m, wgeo(m), N > 0 ? N : 1).total;
}

int wide_forward(const Net& n, const float* params, const float* X, int64_t N, float* Y, float* dY, void* ws,
                 int64_t ws_bytes, hipStream_t s) {
  return run(n, false, nullptr, params, X, N, Y, dY, ws, ws_bytes, s);
}

int wide_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N, void* ws,
              int64_t ws_bytes, hipStream_t s) {
  return run(n, rq.grad != nullptr, &rq, params, X, N, nullptr, nullptr, ws, ws_bytes, s);
}

}  // namespace pinn
