// pinn_wide.hip — host side of the wide MFMA engine (64 < W <= 256): weight packing, chunked
// point loop, per-layer launches.  Kernels: wide_kernel.h.
#include <type_traits>
#include "wide_kernel.h"

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

int cus() { return device_cu_count(); }

constexpr int64_t ACT_BUDGET_BYTES = (int64_t)6 << 30;   // activation workspace per chunk

struct WLayout {
  int64_t chunk_pts, chunk_tiles, n_chunks;
  int64_t wp, wtp, bp, wp16, wtp16, act, act_stride, gA, gB, gZ, gout, sums, total;
  int grid;
};
WLayout wlayout(const Net& n, const WGeo& g, int64_t N) {
  WLayout w;
  const int K1 = 1 + PINN_MAX_DIRS;
  // L jets + 2 adjoint buffers (+1 zbar buffer in bf16 mode, where two waves share a tile)
  const int64_t per_pt = (int64_t)K1 * g.WP * 4 * (n.L + 2 + (n.prec == PINN_PREC_BF16 ? 1 : 0)) + (int64_t)K1 * 16 * 4;
  // whole number of tiles per wave in every full chunk (no tail imbalance): multiple of waves * 16 points
  const int64_t quantum = (int64_t)cus() * WIDE_WAVES * 16;
  int64_t cp = ACT_BUDGET_BYTES / per_pt;
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
  w.act = off; off += w.act_stride * n.L;
  w.gA = off; off += w.act_stride;
  w.gB = off; off += w.act_stride;
  w.gZ = off; off += (n.prec == PINN_PREC_BF16) ? w.act_stride : 0;
  w.gout = off; off += al256(w.chunk_tiles * K1 * 256 * 4);
  w.sums = off; off += al256(w.n_chunks * w.grid * MAX_SUMS * 4);
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
  auto woff = [&](int l) { return l == 0 ? 0 : g.WP * 16 + (l - 1) * g.WP * g.WP; };
  auto act_l = [&](int l) { return (float*)(base + w.act + (int64_t)(l - 1) * w.act_stride); };   // a_l, l = 1..L
  float* gA = (float*)(base + w.gA);
  float* gB = (float*)(base + w.gB);
  float* gZ = (float*)(base + w.gZ);
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
    Lp.sums_slot = (int)(ch * w.grid);
    const int grid = (int)((Lp.n_tiles + WIDE_WAVES - 1) / WIDE_WAVES < w.grid ? (Lp.n_tiles + WIDE_WAVES - 1) / WIDE_WAVES : w.grid);
    // zero this chunk's rows of the partial-sum table (a smaller grid leaves rows untouched)
    if (hipMemsetAsync(P.wg_sums + (int64_t)Lp.sums_slot * MAX_SUMS, 0, (size_t)w.grid * MAX_SUMS * 4, s) != hipSuccess) {
      set_error("hipMemsetAsync failed"); return PINN_ERR_LAUNCH;
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
      float* zdst = (prec == PINN_PREC_BF16) ? gZ : gcur;   // fp32: one wave per tile -> in place
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
                         (int64_t)w.n_chunks * w.grid, 0, rq->n_terms, rq->sums);
    if (P.loss_kind & 2)
      hipLaunchKernelGGL(k_wide_reduce_sums, dim3(rq->n_cols), dim3(256), 0, s, (const float*)P.wg_sums,
                         (int64_t)w.n_chunks * w.grid, MSE_SUM0, rq->n_cols, rq->mse_sums);
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
