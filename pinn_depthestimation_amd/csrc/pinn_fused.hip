// pinn_fused.hip — host side of the fused MFMA engine: weight packing, workspace carve,
// launch geometry, cross-workgroup reductions.  Kernel: fused_kernel.h.
#include <type_traits>
#include "fused_coop_kernel.h"
#include "fused_batch_kernel.h"

namespace pinn {

int launch_fused_drop64(int K1, const FusedParams& P, int grid, size_t lds, hipStream_t s);   // pinn_fused_w64_drop.hip
int launch_fused_plain(int WP, const FusedParams& P, int cus, hipStream_t s);                      // pinn_fused_plain.hip
int64_t fused_plain_min_tiles(int WP, int cus);

namespace {

int padded_width(int W) { return W <= 16 ? 16 : (W <= 32 ? 32 : 64); }

struct Geo {
  int WP, NTH, PW, PB, PP;
  int64_t slot_floats_k4;   // per spilled layer at K1 = 4
};
Geo geo_of(const Net& n) {
  Geo g;
  g.WP = padded_width(n.W); g.NTH = g.WP / 16;
  g.PW = g.WP * 16 + (n.L - 1) * g.WP * g.WP + 16 * g.WP;
  g.PB = n.L * g.WP + 16;
  g.PP = g.PW + g.PB;
  g.slot_floats_k4 = (int64_t)4 * g.NTH * 256;
  return g;
}

int cu_count() { return device_cu_count(); }   // of the CURRENT device (common.h: cached per device)

constexpr int64_t LDS_LIMIT = 160 * 1024;

int64_t lds_fixed_bytes() { return (int64_t)(MAX_LOCKS + FUSED_WAVES * TB_PER_WAVE * TB_FLOATS + FUSED_WAVES * MAX_SUMS) * 4; }
bool fits_lds(const Geo& g) { return (int64_t)g.PP * 4 + lds_fixed_bytes() <= LDS_LIMIT; }

// cooperative (four waves per tile) kernel for small point sets (fused_coop_kernel.h): one workgroup
// per CU runs a tile in ~45 us against ~110 us for k_fused's one-wave tile, so it wins while there is
// at most one tile per CU (measured: loss+grad 127 -> 69 us at N = 243, 137 -> 93 us at N = 4096,
// break-even at N = 8192).  desc.engine = PINN_ENGINE_FUSED_TILE / _COOP forces one of the two (Net::fused_kernel).
int64_t coop_lds_bytes(const Net& n, const Geo& g, bool grad) {
  return ((int64_t)(grad ? g.PP : 0) + 2 * (int64_t)n.K1 * 4 * TB_FLOATS + MAX_SUMS) * 4;
}
bool use_coop(const Net& n, const Geo& g, bool grad, int64_t N) {
  const int forced = n.fused_kernel == FUSED_KERNEL_COOP ? 1 : (n.fused_kernel == FUSED_KERNEL_TILE ? 0 : -1);
  if (forced == 0 || g.WP != 64 || n.drop_p > 0.f) return false;
  if (coop_lds_bytes(n, g, grad) > LDS_LIMIT) return false;
  if (forced == 1) return true;
  return (N + 15) / 16 <= (int64_t)cu_count();
}

// Batch kernel (fused_batch_kernel.h): narrow networks, gradient passes, enough tiles that every wave of the chip
// gets at least one full batch of FUSED_BATCH_T tiles.  desc.engine = PINN_ENGINE_FUSED_BATCH forces it at any N
// (ragged batches are handled: tiles past the end are computed on a clamped point and contribute nothing).
bool batch_supported(const Net& n, const Geo& g) {
  return n.drop_p == 0.f && g.WP <= 32 && n.L >= 1 && n.L + 1 <= MAX_LOCKS && fused_batch_has_kernel(g.WP, n.W, n.d_in, n.K1, n.act);
}
constexpr int64_t BATCH_MIN_TILES = 256;   // AUTO: below this many tiles (4096 points) the tile kernel keeps the request
bool use_batch(const Net& n, const Geo& g, bool grad, int64_t N) {
  if (!grad || !batch_supported(n, g)) return false;
  if (n.fused_kernel == FUSED_KERNEL_BATCH) return true;
  if (n.fused_kernel != FUSED_KERNEL_AUTO) return false;
  return (N + 15) / 16 >= BATCH_MIN_TILES;
}
// tiles per wave and batch: the full batch once every wave of the chip gets one, else ONE tile per wave (the latency of
// a layer step scales with the batch, and a small point set wants its tiles spread over as many waves as there are)
int batch_T_for(const Geo& g, int K1, int64_t n_tiles) {
  const int T = batch_tiles(g.WP, K1);
  return n_tiles >= (int64_t)cu_count() * BATCH_WAVES * batch_occ(g.WP, K1) * T ? T : 1;
}
int64_t batch_lds_fixed_bytes(int WP, int K1) { return (int64_t)(BATCH_WAVES * batch_pads(WP, K1) * TB_FLOATS + BATCH_WAVES * MAX_SUMS) * 4; }
int64_t batch_lds_comb_bytes(int WP) { return (int64_t)batch_comb_floats(WP) * 4; }   // T = 1 + atomic sink: bwgrad_flush_wg's two buffers
int batch_ks(const Net& n) { return n.W <= 12 ? 3 : (n.W <= 16 ? 4 : (n.W <= 20 ? 5 : 8)); }   // k-steps of the kernel instance
int batch_grid(int64_t n_tiles, int T, int occ) {
  const int64_t nb = (n_tiles + T - 1) / T;
  const int64_t want = (nb + BATCH_WAVES - 1) / BATCH_WAVES;
  const int64_t cap = (int64_t)cu_count() * occ;
  return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

int grid_for(int64_t n_tiles, bool one_per_cu, int per_cu = 2) {
  int64_t want = (n_tiles + FUSED_WAVES - 1) / FUSED_WAVES;
  int64_t cap = (int64_t)cu_count() * (one_per_cu ? 1 : per_cu);
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

struct WsLayout {
  int64_t wp, wtp, bp, scratch, wg_sums, wg_grads, total;
  int max_grid;
};
int64_t al(int64_t v) { return (v + 255) & ~(int64_t)255; }

WsLayout ws_layout(const Net& n, const Geo& g, int64_t N) {
  WsLayout w;
  const int64_t n_tiles = (N + 15) / 16;
  w.max_grid = grid_for(n_tiles, false, g.WP == 16 && PINN_FUSED_W16_WAVES > 2 ? PINN_FUSED_W16_WAVES : 2);
  {   // the cooperative kernel launches one workgroup per tile (up to 2 per CU)
    const int64_t cap = 2 * (int64_t)cu_count();
    const int64_t coop_grid = n_tiles < cap ? n_tiles : cap;
    if (coop_grid > w.max_grid) w.max_grid = (int)coop_grid;
  }
  int64_t off = 0;
  w.wp = off; off += al((int64_t)g.PW * 4);
  w.wtp = off; off += al((int64_t)g.PW * 4);
  w.bp = off; off += al((int64_t)g.PB * 4);
  int64_t scratch_bytes = (int64_t)w.max_grid * FUSED_WAVES * n.L * g.slot_floats_k4 * 4;
  if (batch_supported(n, g)) {   // the batch kernel's slots: T tiles x (L - 1) layers x K1 <= 4 x KS x 64 floats per wave
    int64_t b = 0;      // (one workspace serves the K1 = 3 and K1 = 4 instances: the larger of the two)
    for (int k1 = 3; k1 <= 4; ++k1) {
      const int T = batch_T_for(g, k1, n_tiles);
      const int64_t bk = (int64_t)batch_grid(n_tiles, T, batch_occ(g.WP, k1)) * BATCH_WAVES * T * (n.L > 1 ? n.L - 1 : 1) * k1 * batch_ks(n) * 64 * 4;
      if (bk > b) b = bk;
    }
    if (b > scratch_bytes) scratch_bytes = b;
  }
  w.scratch = off; off += al(scratch_bytes);
  w.wg_sums = off; off += al((int64_t)w.max_grid * MAX_SUMS * 4);
  const int64_t copies = w.max_grid;   // one (padded) gradient copy per workgroup, in LDS or — too large for it — here
  w.wg_grads = off; off += al(copies * g.PP * 4);
  w.total = off;
  return w;
}

// Which real unit sits at padded index j.  perm = 0: j itself.  perm = 1 (k_fused_batch, fused_batch_kernel.h): hidden
// units and network inputs in k-step-major order, j <-> 16*(j/16) + perm16(j%16) (an involution, so the same formula
// maps a real unit to its padded index); network outputs always stay in natural order.
__host__ __device__ inline int unit_at(int j, bool permuted) { return permuted ? (j & ~15) + perm16(j & 15) : j; }
// packing flags: PACK_PERM = the batch kernel's order (hidden units and inputs permuted, outputs natural); PACK_RMAJOR =
// register-major gradient blocks (reduction kernels only); PACK_IN / PACK_OUT = ONLY the network inputs / outputs
// permuted (k_fused<..., KRO > 0>: one k-step for the first layer, ceil(d_out / 4) for the output layer's reverse GEMM)
constexpr int PACK_PERM = 1, PACK_RMAJOR = 2, PACK_IN = 4, PACK_OUT = 8;
__host__ __device__ inline bool row_permuted(int flags, int l, int L) { return l < L ? (flags & PACK_PERM) != 0 : (flags & PACK_OUT) != 0; }
__host__ __device__ inline bool col_permuted(int flags, int l) { return (flags & PACK_PERM) != 0 || (l == 0 && (flags & PACK_IN) != 0); }

// flat torch-layout parameters -> padded row-major W, padded transposed W, padded bias
__global__ void k_pack(Net n, int WP, const float* __restrict__ params, float* __restrict__ Wp,
                       float* __restrict__ WTp, float* __restrict__ Bp, int PW, int PB, int flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < PW) {
    // which layer block does padded index i fall in?
    int l, rem;
    if (i < WP * 16) { l = 0; rem = i; }
    else {
      const int j = i - WP * 16;
      l = 1 + j / (WP * WP); rem = j % (WP * WP);
      if (l > n.L) { l = n.L; rem = i - (WP * 16 + (n.L - 1) * WP * WP); }
    }
    const int inP = (l == 0) ? 16 : WP, outP = (l == n.L) ? 16 : WP;
    const int in_d = n.in_dim(l), out_d = n.out_dim(l);
    const int base = i - rem;
    {  // row-major [out][in]
      const int o = unit_at(rem / inP, row_permuted(flags, l, n.L)), c = unit_at(rem % inP, col_permuted(flags, l));
      Wp[i] = (o < out_d && c < in_d) ? params[n.w_off(l) + (int64_t)o * in_d + c] : 0.f;
    }
    {  // transposed [in][out]
      const int c = unit_at(rem / outP, col_permuted(flags, l)), o = unit_at(rem % outP, row_permuted(flags, l, n.L));
      WTp[base + rem] = (o < out_d && c < in_d) ? params[n.w_off(l) + (int64_t)o * in_d + c] : 0.f;
    }
  }
  if (i < PB) {
    int l = i / WP, o = i % WP;
    if (l >= n.L) { l = n.L; o = i - n.L * WP; }
    o = unit_at(o, row_permuted(flags, l, n.L));
    Bp[i] = (o < n.out_dim(l)) ? params[n.b_off(l) + o] : 0.f;
  }
}

__global__ void k_reduce_sums(const float* __restrict__ wg_sums, int grid, int col0, int nt, float* __restrict__ out) {
  const int t = col0 + blockIdx.x;
  __shared__ double red[256];
  double v = 0.0;
  for (int b = threadIdx.x; b < grid; b += 256) v += (double)wg_sums[(int64_t)b * MAX_SUMS + t];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x < nt) out[blockIdx.x] = (float)red[0];
}

// grad_flat[real index] += sum over copies of the padded per-workgroup gradients.  64 parameters x
// 4 copy groups per block; each thread adds its group's copies in index order and the 4 partial sums
// are combined in a fixed order, so the result does not depend on scheduling.
// a thread's share of the copies, added in index order; the loads of eight copies are issued together (one memory
// round trip per eight instead of per copy: 59 -> 12 us on the 768 copies of the 10x10 net) — the ORDER of the
// additions, and with it the result, is unchanged
__device__ __forceinline__ float sum_copies(const float* __restrict__ wg, int64_t PP, int pidx, int c0, int c1) {
  float s = 0.f;
  int c = c0;
  for (; c + 8 <= c1; c += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = wg[(int64_t)(c + j) * PP + pidx];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; c < c1; ++c) s += wg[(int64_t)c * PP + pidx];
  return s;
}

__global__ void k_reduce_grads(Net n, int WP, const float* __restrict__ wg, int copies, int PP, int PW,
                               float* __restrict__ grad, int flags) {
  const int rmajor = (flags & PACK_RMAJOR) != 0;   // [r][lane] inside a 16x16 block (bwgrad_flush, BSINK_ATOMIC)
  __shared__ float part[4][64];
  const int lane_p = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane_p;
  float s = 0.f;
  if (i < n.n_params()) {
    const int l = n.layer_of(i);
    const int64_t off = n.w_off(l);
    const int64_t r = i - off;
    const int in_d = n.in_dim(l), out_d = n.out_dim(l);
    const int inP = (l == 0) ? 16 : WP;
    const int wo = (l == 0) ? 0 : WP * 16 + (l - 1) * WP * WP;
    int pidx;
    if (r < (int64_t)in_d * out_d) {   // fragment-native block layout (fused_kernel.h, GradSink)
      const int row = unit_at((int)(r / in_d), row_permuted(flags, l, n.L)), col = unit_at((int)(r % in_d), col_permuted(flags, l)), ntn = inP / 16;
      const int blk = (row >> 4) * ntn + (col >> 4), ln = ((row & 15) >> 2) * 16 + (col & 15);
      pidx = rmajor ? wo + (blk * 4 + (row & 3)) * 64 + ln : wo + (blk * 64 + ln) * 4 + (row & 3);
    } else pidx = PW + l * WP + unit_at((int)(r - (int64_t)in_d * out_d), row_permuted(flags, l, n.L));
    const int per = (copies + 3) / 4;
    const int c0 = grp * per, c1 = (c0 + per < copies) ? c0 + per : copies;
    s = sum_copies(wg, PP, pidx, c0, c1);
  }
  part[grp][lane_p] = s;
  __syncthreads();
  if (grp == 0 && i < n.n_params()) grad[i] += (part[0][lane_p] + part[1][lane_p]) + (part[2][lane_p] + part[3][lane_p]);
}

// One kernel for everything that follows the fused pass in an Adam iteration (pinn_loss_grad_adam_step): the
// gradient reduction of k_reduce_grads (same partial sums, same order: bit-identical), torch.optim.Adam's update
// (k_adam's arithmetic, pinn_abi.hip) on the parameter it just summed, the refreshed entries of the packed W / W^T / b
// the next pass reads (k_pack's mapping, inverted: padding entries stay zero), and — last block — the loss sums of
// k_reduce_sums.  Five launches of ~4.7 us each become one at the reference's own problem sizes (N_res = 243).
__global__ void k_finish_adam(Net n, int WP, const float* __restrict__ wg, int copies, int PP, int PW,
                              const float* __restrict__ wg_sums, int grid, int n_terms, float* __restrict__ term_sums,
                              int n_cols, float* __restrict__ col_sums, float* __restrict__ grad,
                              float* __restrict__ params, float* __restrict__ m, float* __restrict__ v,
                              float* __restrict__ Wp, float* __restrict__ WTp, float* __restrict__ Bp,
                              float w1, float b2, float w2, float eps, float step_size, float bc2_sqrt,
                              int n_loss_rows, const float* __restrict__ loss_rows, float* __restrict__ losses,
                              int row_cols, int flags) {
#pragma clang fp contract(off)
  const int rmajor = (flags & PACK_RMAJOR) != 0;
  if (blockIdx.x == gridDim.x - 1) {          // loss sums: double, fixed order (k_reduce_sums)
    __shared__ double red[256];
    // [col sums (row_cols of them) | term sums], for the optional weighted losses.  row_cols is the CALLER's column
    // count — the stride of loss_rows (pinn_hip.h) — also when this pass carried no fidelity columns (n_cols = 0:
    // a residual-only request with n_res == N): their sums are then zeros, not a shifted layout.
    __shared__ double ssum[2 * PINN_MAX_ROLES + 8];
    if (threadIdx.x < 2 * PINN_MAX_ROLES + 8) ssum[threadIdx.x] = 0.0;
    __syncthreads();
    for (int j = 0; j < n_terms + n_cols; ++j) {
      const int t = j < n_terms ? j : MSE_SUM0 + (j - n_terms);
      double a = 0.0;
      for (int b = threadIdx.x; b < grid; b += 256) a += (double)wg_sums[(int64_t)b * MAX_SUMS + t];
      red[threadIdx.x] = a;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        if (j < n_terms) { term_sums[j] = (float)red[0]; ssum[row_cols + j] = (double)(float)red[0]; }
        else { col_sums[j - n_terms] = (float)red[0]; ssum[j - n_terms] = (double)(float)red[0]; }
      }
      __syncthreads();
    }
    if ((int)threadIdx.x < n_loss_rows) {
      double a = 0.0;
      for (int j = 0; j < row_cols + n_terms; ++j) a += (double)loss_rows[threadIdx.x * (row_cols + n_terms) + j] * ssum[j];
      losses[threadIdx.x] = (float)a;
    }
    return;
  }
  __shared__ float part[4][64];
  const int lane_p = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane_p;
  float s = 0.f;
  int l = 0, row = 0, col = 0, wo = 0, in_d = 1, out_d = 1;
  bool is_w = false;
  if (i < n.n_params()) {
    l = n.layer_of(i);
    const int64_t r = i - n.w_off(l);
    in_d = n.in_dim(l); out_d = n.out_dim(l);
    const int inP = (l == 0) ? 16 : WP;
    wo = (l == 0) ? 0 : WP * 16 + (l - 1) * WP * WP;
    int pidx;
    is_w = r < (int64_t)in_d * out_d;
    if (is_w) {   // fragment-native block layout (fused_kernel.h, GradSink); row / col: PADDED indices from here on
      row = unit_at((int)(r / in_d), row_permuted(flags, l, n.L)); col = unit_at((int)(r % in_d), col_permuted(flags, l));
      const int ntn = inP / 16;
      const int blk = (row >> 4) * ntn + (col >> 4), ln = ((row & 15) >> 2) * 16 + (col & 15);
      pidx = rmajor ? wo + (blk * 4 + (row & 3)) * 64 + ln : wo + (blk * 64 + ln) * 4 + (row & 3);
    } else { row = unit_at((int)(r - (int64_t)in_d * out_d), row_permuted(flags, l, n.L)); pidx = PW + l * WP + row; }
    const int per = (copies + 3) / 4;
    const int c0 = grp * per, c1 = (c0 + per < copies) ? c0 + per : copies;
    s = sum_copies(wg, PP, pidx, c0, c1);
  }
  part[grp][lane_p] = s;
  __syncthreads();
  if (grp == 0 && i < n.n_params()) {
    const float gi = (part[0][lane_p] + part[1][lane_p]) + (part[2][lane_p] + part[3][lane_p]);
    grad[i] = gi;
    const float mi = m[i] + w1 * (gi - m[i]);
    float vi = v[i] * b2;
    vi = vi + (w2 * gi) * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = params[i] - step_size * (mi / denom);
    params[i] = pn;
    if (is_w) {
      const int inP = (l == 0) ? 16 : WP, outP = (l == n.L) ? 16 : WP;
      Wp[wo + row * inP + col] = pn;
      WTp[wo + col * outP + row] = pn;
    } else Bp[l * WP + row] = pn;
  }
}

int run(const Net& n, bool grad, const LossReq* rq, const float* params, const float* X, int64_t N, float* Y,
        float* dY, void* ws, int64_t ws_bytes, hipStream_t s) {
  const Geo g = geo_of(n);
  const WsLayout w = ws_layout(n, g, N);
  if (!ws || ws_bytes < w.total) {
    set_error("workspace too small: need %lld bytes, got %lld", (long long)w.total, (long long)ws_bytes);
    return PINN_ERR_WORKSPACE;
  }
  char* base = (char*)ws;
  FusedParams P;
  memset(&P, 0, sizeof(P));
  P.d_in = n.d_in; P.d_out = n.d_out; P.L = n.L; P.act = n.act;
  for (int j = 0; j < PINN_MAX_DIRS; ++j) P.dir_col[j] = n.dir_col[j];
  P.N = N; P.n_tiles = (N + 15) / 16;
  P.n_split = rq ? rq->n_split : -1;
  P.X = X;
  P.Wp = (const float*)(base + w.wp); P.WTp = (const float*)(base + w.wtp); P.Bp = (const float*)(base + w.bp);
  P.scratch = (float*)(base + w.scratch);
  P.scratch_per_wave = (int64_t)n.L * n.K1 * g.NTH * 256;
  P.Y = Y; P.dY = dY;
  P.wg_sums = (float*)(base + w.wg_sums);
  P.wg_grads = (float*)(base + w.wg_grads);
  P.PW = g.PW; P.PB = g.PB;
  P.acc_lds = (grad && fits_lds(g)) ? 1 : 0;
  P.lds_acc_floats = P.acc_lds ? g.PP : 0;
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
  const bool batch = use_batch(n, g, grad, N);
  const bool coop = !batch && use_coop(n, g, grad, N);
  if (coop) {
    P.acc_lds = grad ? 1 : 0;
    P.lds_acc_floats = grad ? g.PP : 0;
  }
  if (batch) {   // the batch kernel carves its own pads (batch_pads per wave); the gradient copy stays in LDS if it still fits
    // a gradient copy per wave in LDS if four of them fit, else atomics into BATCH_ATOMIC_COPIES shared copies (bwgrad_flush)
    P.acc_lds = ((int64_t)BATCH_WAVES * g.PP * 4 + batch_lds_fixed_bytes(g.WP, n.K1)) * batch_occ(g.WP, n.K1) <= LDS_LIMIT ? 1 : 0;
    P.lds_acc_floats = P.acc_lds ? BATCH_WAVES * g.PP : 0;
  }
  const size_t lds = batch ? (size_t)P.lds_acc_floats * 4 + (size_t)batch_lds_fixed_bytes(g.WP, n.K1) +
                                 (size_t)(!P.acc_lds && batch_T_for(g, n.K1, P.n_tiles) == 1 ? batch_lds_comb_bytes(g.WP) : 0)
                   : coop ? (size_t)coop_lds_bytes(n, g, grad) : (size_t)P.lds_acc_floats * 4 + (size_t)lds_fixed_bytes();
  // 8x64 gradient kernels fill the register file and most of LDS (1 workgroup per CU); the narrow
  // networks' kernels fit 2 waves per SIMD, which hides their per-layer latencies
  const bool one_per_cu = grad && P.acc_lds && !(g.WP <= 32 && 2 * (int64_t)lds <= LDS_LIMIT);
  int grid = grid_for(P.n_tiles, one_per_cu, g.WP == 16 && PINN_FUSED_W16_WAVES > 2 && (int64_t)PINN_FUSED_W16_WAVES * (int64_t)lds <= LDS_LIMIT ? PINN_FUSED_W16_WAVES : 2);
  if (coop) {   // one workgroup per tile, at most one per CU (gradient kernels fill the LDS)
    const int64_t cap = (int64_t)cu_count() * (grad ? 1 : 2);
    grid = (int)(P.n_tiles < cap ? (P.n_tiles < 1 ? 1 : P.n_tiles) : cap);
  }

  if (batch) {   // one workgroup per CU, one wave per SIMD; slots of T tiles x (L - 1) layers per wave
    const int T = batch_T_for(g, n.K1, P.n_tiles);
    P.batch_T = T;
    grid = batch_grid(P.n_tiles, T, batch_occ(g.WP, n.K1));
    P.scratch_per_wave = (int64_t)T * (n.L > 1 ? n.L - 1 : 1) * n.K1 * batch_ks(n) * 64;
  }
  // k_fused<64, ..., KRO > 0>: the specialised-epilogue kernels take inputs / outputs in k-step-major order
  if (!batch && !coop && n.drop_p == 0.f && g.WP == 64 && grad && P.acc_lds && n.act == PINN_ACT_TANH && P.loss_kind == 1 && !Y && P.n_split < 0 &&
      n.d_in <= 4) {
    const int rid = P.residual_id;
    P.io1 = (n.K1 == 4 && rid == PINN_RES_NAVIER_STOKES && n.d_out <= 4) ||
            (n.K1 == 3 && rid == PINN_RES_PHYSICS_EQUATION && n.d_out <= 8) ||
            (n.K1 == 3 && (rid == PINN_RES_CONTINUITY_ONLY || rid == PINN_RES_CONTINUITY_FTEMP) && n.d_out <= 4);
    if (P.io1) {   // the epilogue addresses the output tile / the input jet by PADDED index
      for (int j = 0; j < PINN_MAX_ROLES; ++j) if (P.out_col[j] >= 0) P.out_col[j] = perm16(P.out_col[j]);
      for (int j = 0; j < PINN_MAX_DIRS; ++j) if (P.dir_col[j] >= 0) P.dir_col[j] = perm16(P.dir_col[j]);
    }
  }
  const int perm = (batch ? PACK_PERM : 0) | (P.io1 ? PACK_IN | PACK_OUT : 0);   // unit order the pass expects

  const AdamReq* adam = rq ? rq->adam : nullptr;
  const int packN = g.PW > g.PB ? g.PW : g.PB;
  if (!(adam && adam->packed_valid))
    hipLaunchKernelGGL(k_pack, dim3((packN + 255) / 256), dim3(256), 0, s, n, g.WP, params, (float*)(base + w.wp),
                       (float*)(base + w.wtp), (float*)(base + w.bp), g.PW, g.PB, perm);
  // gradient copies the reduction reads: one per workgroup, or the batch kernel's shared atomic copies (register-major)
  const int rmajor = batch && !P.acc_lds ? 1 : 0;
  const int n_copies = rmajor ? (grid < BATCH_ATOMIC_COPIES ? grid : BATCH_ATOMIC_COPIES) : grid;
  if (grad && !P.acc_lds) {   // global gradient copies start from zero
    if (hipMemsetAsync(P.wg_grads, 0, (size_t)n_copies * g.PP * 4, s) != hipSuccess) {
      set_error("hipMemsetAsync failed"); return PINN_ERR_LAUNCH;
    }
  }
  int rc;
  if (n.drop_p > 0.f) {     // training-mode dropout: its own instances of the tile kernel (pinn_fused_w64_drop.hip)
    P.drop_seed = n.drop_seed; P.drop_thresh = n.drop_thresh; P.drop_scale = 1.f / (1.f - n.drop_p); P.drop_keep = 1.f - n.drop_p;
    if (!(grad && g.WP == 64 && P.acc_lds && n.act == PINN_ACT_TANH)) { set_error("fused engine: no dropout kernel for this request"); return PINN_ERR_UNSUPPORTED; }
    rc = launch_fused_drop64(n.K1, P, grid, lds, s);
  }
  else if (!rq && !grad && !coop && !batch && n.K1 == 1 && Y && !dY && n.fused_kernel == FUSED_KERNEL_AUTO &&
           P.n_tiles >= fused_plain_min_tiles(g.WP, cu_count())) {
    // pinn_forward on enough points to give every wave the chip holds a pass of four tiles: the plain forward's own
    // kernel, one weight fetch per 64 points (pinn_fused_plain.hip)
    rc = launch_fused_plain(g.WP, P, cu_count(), s);
  }
  else if (batch) rc = g.WP == 16 ? launch_fused_batch<16>(n.W, n.d_in, n.K1, P, grid, lds, s)
                             : launch_fused_batch<32>(n.W, n.d_in, n.K1, P, grid, lds, s);
  else if (coop) rc = launch_fused_coop(n.K1, grad, P, grid, lds, s);
  else switch (g.WP) {
    case 16: rc = launch_fused<16>(n.K1, grad, P, grid, lds, s); break;
    case 32: rc = launch_fused<32>(n.K1, grad, P, grid, lds, s); break;
    default: rc = launch_fused<64>(n.K1, grad, P, grid, lds, s); break;
  }
  if (rc) return rc;
  if (adam) {
    const int64_t np = n.n_params();
    hipLaunchKernelGGL(k_finish_adam, dim3((unsigned)((np + 63) / 64) + 1), dim3(256), 0, s, n, g.WP,
                       (const float*)P.wg_grads, n_copies, g.PP, g.PW, (const float*)P.wg_sums, grid,
                       (P.loss_kind & 1) ? rq->n_terms : 0, rq->sums, (P.loss_kind & 2) ? rq->n_cols : 0, rq->mse_sums,
                       rq->grad, adam->params, adam->m, adam->v, (float*)(base + w.wp), (float*)(base + w.wtp),
                       (float*)(base + w.bp), adam->w1, adam->b2, adam->w2, adam->eps, adam->step_size, adam->bc2_sqrt,
                       adam->n_loss_rows, adam->loss_rows, adam->losses, rq->n_cols, perm | (rmajor ? PACK_RMAJOR : 0));
    return check_launch("fused finish + adam");
  }
  if (rq) {
    if (P.loss_kind & 1)
      hipLaunchKernelGGL(k_reduce_sums, dim3(rq->n_terms), dim3(256), 0, s, (const float*)P.wg_sums, grid, 0,
                         rq->n_terms, rq->sums);
    if (P.loss_kind & 2)
      hipLaunchKernelGGL(k_reduce_sums, dim3(rq->n_cols), dim3(256), 0, s, (const float*)P.wg_sums, grid, MSE_SUM0,
                         rq->n_cols, rq->mse_sums);
    if (grad) {
      const int copies = n_copies;
      const int64_t np = n.n_params();
      hipLaunchKernelGGL(k_reduce_grads, dim3((unsigned)((np + 63) / 64)), dim3(256), 0, s, n, g.WP,
                         (const float*)P.wg_grads, copies, g.PP, g.PW, rq->grad, perm | (rmajor ? PACK_RMAJOR : 0));
    }
  }
  return check_launch("fused reductions");
}

}  // namespace

// batch kernel instances: pinn_fused_batch_w{16,32}_k{3,4}.hip
template <int WP, int K1>
int launch_fused_batch_k(int W, int d_in, const FusedParams& P, int grid, size_t lds, hipStream_t s);
template <int WP>
int launch_fused_batch(int W, int d_in, int K1, const FusedParams& P, int grid, size_t lds, hipStream_t s) {
  return K1 == 3 ? launch_fused_batch_k<WP, 3>(W, d_in, P, grid, lds, s) : launch_fused_batch_k<WP, 4>(W, d_in, P, grid, lds, s);
}
template int launch_fused_batch<16>(int, int, int, const FusedParams&, int, size_t, hipStream_t);
template int launch_fused_batch<32>(int, int, int, const FusedParams&, int, size_t, hipStream_t);
bool fused_batch_has_kernel(int WP, int W, int d_in, int K1, int act) {
  return (WP == 16 || WP == 32) && W >= 1 && d_in <= 8 && (K1 == 3 || K1 == 4) && act == PINN_ACT_TANH;
}

bool fused_supports(const Net& n, bool want_grad) {
  if (n.L + 1 > MAX_LOCKS) return false;
  if (n.drop_p > 0.f) {   // dropout: gradient passes of tanh networks of padded width 64 whose gradient copy fits LDS
    if (!(want_grad && padded_width(n.W) == 64 && n.act == PINN_ACT_TANH && fits_lds(geo_of(n)) && n.fused_kernel != FUSED_KERNEL_COOP &&
          (n.K1 == 1 || n.K1 == 3 || n.K1 == 4)))
      return false;
  }
  if (want_grad && n.K1 == 2) return false;   // no k = 1 gradient kernels (no residual of the reference has one direction)
  if (n.fused_kernel == FUSED_KERNEL_COOP && padded_width(n.W) != 64) return false;
  return n.W <= 64 && n.d_in <= 16 && n.d_out <= 16 && n.L >= 1 && n.K1 >= 1 && n.K1 <= 4;
}

// The folded update needs the whole request in ONE pass (the split request on the width-64 tile kernel runs as two).
bool fused_supports_adam(const Net& n, const LossReq& rq, int64_t N) {
  if (!rq.grad || !fused_supports(n, true)) return false;
  if (rq.kind == 2 && rq.n_split >= 0) {
    const Geo g = geo_of(n);
    if (g.WP == 64 && !use_coop(n, g, true, N)) return false;
  }
  return true;
}

int64_t fused_workspace_bytes(const Net& n, int64_t N) {
  if (!fused_supports(n, false) && !fused_supports(n, true)) return -1;   // (dropout: gradient passes only)
  return ws_layout(n, geo_of(n), N > 0 ? N : 1).total;
}

int fused_forward(const Net& n, const float* params, const float* X, int64_t N, float* Y, float* dY, void* ws,
                  int64_t ws_bytes, hipStream_t s) {
  return run(n, false, nullptr, params, X, N, Y, dY, ws, ws_bytes, s);
}

int fused_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N, void* ws,
               int64_t ws_bytes, hipStream_t s) {
  const bool grad = rq.grad != nullptr;
  if (rq.kind == 2 && rq.n_split >= 0) {
    // Split mode lives in the cooperative kernel and the narrow (WP < 64) tile kernels only.  A
    // request that would run on the width-64 tile kernel (large N, where a second launch is noise)
    // is served as two passes on the same stream: residual on the collocation points, then the
    // fidelity columns on the rest with the k = 0 network.
    const Geo g = geo_of(n);
    if (g.WP == 64 && !use_coop(n, g, grad, N)) {
      LossReq r0 = rq; r0.kind = 0; r0.n_split = -1;
      int rc = PINN_OK;
      if (rq.n_split > 0) rc = fused_loss(n, r0, params, X, rq.n_split, ws, ws_bytes, s);
      else (void)hipMemsetAsync(rq.sums, 0, rq.n_terms * sizeof(float), s);
      if (rc) return rc;
      LossReq r1 = rq; r1.kind = 1; r1.n_split = -1;
      Net n1 = n; n1.k = 0; n1.K1 = 1;
      return fused_loss(n1, r1, params, X + rq.n_split * n.d_in, N - rq.n_split, ws, ws_bytes, s);
    }
  }
  if (!fused_supports(n, grad)) { set_error("fused engine: no kernel for this request (k = %d, gradient %d)", n.k, (int)grad); return PINN_ERR_UNSUPPORTED; }
  return run(n, grad, &rq, params, X, N, nullptr, nullptr, ws, ws_bytes, s);
}

}  // namespace pinn
