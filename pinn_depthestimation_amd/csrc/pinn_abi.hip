// pinn_abi.hip — extern "C" entry points of libpinn_hip.so (see include/pinn_hip.h)
// and the engine dispatch.  No torch types, no exceptions, no allocation.
#include <string.h>
#include <mutex>
#include <utility>
#include <vector>
#include "common.h"

namespace pinn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// The only process-wide state of the library: immutable facts about devices / kernels, filled on first use.
namespace {
std::mutex g_cache_mu;
constexpr int MAX_DEVICES = 64;
int g_cus[MAX_DEVICES];                                              // 0 = not yet asked
struct LdsKey { const void* fn; int dev; size_t bytes; };
std::vector<LdsKey> g_lds_set;
}  // namespace

int device_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  std::lock_guard<std::mutex> lk(g_cache_mu);
  if (g_cus[dev] == 0) {
    int v = 0;
    g_cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return g_cus[dev];
}

int ensure_dynamic_lds(const void* kernel, size_t lds_bytes) {
  if (lds_bytes <= 64 * 1024) return PINN_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return PINN_ERR_LAUNCH; }
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (LdsKey& k : g_lds_set)
    if (k.fn == kernel && k.dev == dev) {
      if (k.bytes >= lds_bytes) return PINN_OK;
      hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return PINN_ERR_LAUNCH; }
      k.bytes = lds_bytes;
      return PINN_OK;
    }
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds_bytes, hipGetErrorString(e)); return PINN_ERR_LAUNCH; }
  g_lds_set.push_back(LdsKey{kernel, dev, lds_bytes});
  return PINN_OK;
}

int make_net(const pinn_desc* d, Net* n) {
  if (!d) { set_error("desc is NULL"); return PINN_ERR_INVALID; }
  if (d->d_in < 1 || d->d_out < 1 || d->n_hidden < 1 || d->width < 1) {
    set_error("bad network shape d_in=%d d_out=%d hidden=%d width=%d", d->d_in, d->d_out, d->n_hidden, d->width);
    return PINN_ERR_INVALID;
  }
  if (d->k < 0 || d->k > PINN_MAX_DIRS) { set_error("k=%d outside 0..%d", d->k, PINN_MAX_DIRS); return PINN_ERR_INVALID; }
  if (d->activation != PINN_ACT_TANH && d->activation != PINN_ACT_LEAKY_RELU) {
    // mirrors the ValueError of dnn.py:23 for an unknown init_type
    set_error("invalid activation %d (0 = tanh/'xavier', 1 = leaky_relu/'kaiming')", d->activation);
    return PINN_ERR_INVALID;
  }
  n->d_in = d->d_in; n->d_out = d->d_out; n->L = d->n_hidden; n->W = d->width;
  n->k = d->k; n->K1 = 1 + d->k; n->act = d->activation; n->n_lin = d->n_hidden + 1;
  if (d->precision != PINN_PREC_F32 && d->precision != PINN_PREC_BF16) {
    set_error("invalid precision %d", d->precision); return PINN_ERR_INVALID;
  }
  n->prec = d->precision;
  if (d->engine < PINN_ENGINE_AUTO || d->engine > PINN_ENGINE_FUSED_BATCH) { set_error("invalid engine %d", d->engine); return PINN_ERR_INVALID; }
  n->fused_kernel = d->engine == PINN_ENGINE_FUSED_TILE ? FUSED_KERNEL_TILE
                  : d->engine == PINN_ENGINE_FUSED_COOP ? FUSED_KERNEL_COOP
                  : d->engine == PINN_ENGINE_FUSED_BATCH ? FUSED_KERNEL_BATCH : FUSED_KERNEL_AUTO;
  if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) { set_error("dropout_p=%g outside [0, 1)", (double)d->dropout_p); return PINN_ERR_INVALID; }
  n->drop_p = d->dropout_p; n->drop_seed = d->dropout_seed; n->drop_thresh = dropout_threshold(d->dropout_p);
  if (n->drop_p > 0.f && n->drop_thresh == 0) n->drop_thresh = 1;   // (0 means "off" in the kernels)
  for (int j = 0; j < PINN_MAX_DIRS; ++j) {
    n->dir_col[j] = j < d->k ? d->dir_col[j] : -1;
    if (j < d->k && (d->dir_col[j] < 0 || d->dir_col[j] >= d->d_in)) {
      set_error("dir_col[%d]=%d outside the %d input columns", j, d->dir_col[j], d->d_in);
      return PINN_ERR_INVALID;
    }
  }
  return PINN_OK;
}

// 1 = generic, 2 = fused, 3 = wide
static int pick_engine(const pinn_desc* d, const Net& n, bool want_grad, int* rc) {
  *rc = PINN_OK;
  const int asked = (d->engine == PINN_ENGINE_FUSED_TILE || d->engine == PINN_ENGINE_FUSED_COOP || d->engine == PINN_ENGINE_FUSED_BATCH) ? PINN_ENGINE_FUSED : d->engine;
  if (n.drop_p > 0.f) {     // training-mode dropout: the fused tile kernel for gradient passes at padded width 64
                            // (pinn_fused_w64_drop.hip), the generic engine's kernels for everything else
    if (n.prec != PINN_PREC_F32) { set_error("dropout_p > 0 is implemented in fp32 only"); *rc = PINN_ERR_UNSUPPORTED; return PINN_ENGINE_GENERIC; }
    const bool fused_ok = fused_supports(n, want_grad);
    if (asked == PINN_ENGINE_FUSED && !fused_ok) {
      set_error("dropout_p > 0 on the fused engine: gradient passes of tanh networks of hidden width 33..64 only; this request "
                "runs on the generic engine (engine AUTO or GENERIC)");
      *rc = PINN_ERR_UNSUPPORTED;
    } else if (asked == PINN_ENGINE_WIDE) {
      set_error("dropout_p > 0 runs on the generic engine (engine AUTO or GENERIC), not on engine %d", d->engine);
      *rc = PINN_ERR_UNSUPPORTED;
    }
    return (asked == PINN_ENGINE_AUTO || asked == PINN_ENGINE_FUSED) && fused_ok ? PINN_ENGINE_FUSED : PINN_ENGINE_GENERIC;
  }
  if (n.prec == PINN_PREC_BF16) {   // bf16 operands exist on the wide engine only
    if ((asked != PINN_ENGINE_AUTO && asked != PINN_ENGINE_WIDE) || !wide_supports(n)) {
      set_error("precision bf16 is implemented on the wide engine (64 < width <= 256, tanh, k in {0,2,3}) only");
      *rc = PINN_ERR_UNSUPPORTED;
    }
    return PINN_ENGINE_WIDE;
  }
  if (asked == PINN_ENGINE_GENERIC) return PINN_ENGINE_GENERIC;
  if (asked == PINN_ENGINE_FUSED || asked == PINN_ENGINE_WIDE) {
    const bool ok = asked == PINN_ENGINE_FUSED ? fused_supports(n, want_grad) : wide_supports(n);
    if (!ok) {
      set_error("%s engine does not support this request (width %d, d_in %d, d_out %d, k %d, act %d, gradient %d)",
                asked == PINN_ENGINE_FUSED ? "fused" : "wide", n.W, n.d_in, n.d_out, n.k, n.act, (int)want_grad);
      *rc = PINN_ERR_UNSUPPORTED;
    }
    return asked;
  }
  if (fused_supports(n, want_grad)) return PINN_ENGINE_FUSED;
  if (wide_supports(n)) return PINN_ENGINE_WIDE;
  return PINN_ENGINE_GENERIC;
}

static int residual_terms(int id) {
  switch (id) {
    case PINN_RES_NAVIER_STOKES: return PINN_NS_TERMS;
    case PINN_RES_PHYSICS_EQUATION: return PINN_PE_TERMS;
    case PINN_RES_CONTINUITY_FTEMP: return PINN_CF_TERMS;
    case PINN_RES_CONTINUITY_ONLY: return PINN_CO_TERMS;
  }
  return -1;
}

// Checks the roles the residual USES (its first nr output roles, nd direction roles) and returns in *norm a copy whose
// unused entries are inert: out_col = -1 (matches no output column), dir_of = -1 (quantity 0 is never a direction).
// The engines' per-lane scatter tables are built by searching ALL PINN_MAX_ROLES entries for "which role lives in
// this column"; an unused entry left at 0 (zero-initialised structs: every caller) claimed output column 0 whenever no
// real role sat there, and the output adjoint of that column was read from beyond the roles' rows.
static int check_spec(const Net& n, const pinn_residual_spec* sp, pinn_residual_spec* norm) {
  if (!sp) { set_error("spec is NULL"); return PINN_ERR_INVALID; }
  int nr = 0, nd = 0;
  switch (sp->residual_id) {
    case PINN_RES_NAVIER_STOKES: nr = 4; nd = 3; break;
    case PINN_RES_PHYSICS_EQUATION: nr = 6; nd = 2; break;
    case PINN_RES_CONTINUITY_FTEMP:
    case PINN_RES_CONTINUITY_ONLY: nr = 3; nd = 2; break;
    default: set_error("unknown residual_id %d", sp->residual_id); return PINN_ERR_INVALID;
  }
  for (int r = 0; r < nr; ++r)
    if (sp->out_col[r] < 0 || sp->out_col[r] >= n.d_out) {
      set_error("out_col[%d]=%d outside the %d output columns", r, sp->out_col[r], n.d_out);
      return PINN_ERR_INVALID;
    }
  for (int d = 0; d < nd; ++d)
    if (sp->dir_of[d] < 0 || sp->dir_of[d] >= n.k) {
      set_error("dir_of[%d]=%d but the network carries %d tangent directions", d, sp->dir_of[d], n.k);
      return PINN_ERR_INVALID;
    }
  *norm = *sp;
  for (int r = nr; r < PINN_MAX_ROLES; ++r) norm->out_col[r] = -1;
  for (int d = nd; d < PINN_MAX_DIRS; ++d) norm->dir_of[d] = -1;
  return PINN_OK;
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t P, float w1, float b2, float w2, float eps, float step_size,
                       float bc2_sqrt) {
  // torch.optim.Adam, _single_tensor_adam (the path train.py:192 takes on CPU):
  //   exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2)
  //   denom = (exp_avg_sq.sqrt() / sqrt(bc2)).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr/bc1)
  // Scalars are formed in double on the host exactly as Python does, then cast to
  // fp32 once; contraction is off so every op rounds where torch's rounds.
#pragma clang fp contract(off)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float gi = g[i];
  const float mi = m[i] + w1 * (gi - m[i]);
  float vi = v[i] * b2;
  vi = vi + (w2 * gi) * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

}  // namespace pinn

using namespace pinn;

extern "C" {

int32_t pinn_version(void) { return PINN_ABI_VERSION; }
int32_t pinn_dropout_keep(uint32_t seed, int32_t layer, int32_t feature, int64_t point, float p) {
  return dropout_bits(seed, (uint32_t)layer, (uint32_t)feature, (uint64_t)point) >= dropout_threshold(p) ? 1 : 0;
}
const char* pinn_last_error(void) { return g_err; }

int32_t pinn_param_count(const pinn_desc* desc, int64_t* count) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  if (!count) { set_error("count is NULL"); return PINN_ERR_INVALID; }
  *count = n.n_params();
  return PINN_OK;
}

int32_t pinn_query_workspace(const pinn_desc* desc, int64_t N, int64_t* bytes) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  if (!bytes || N < 0) { set_error("bad arguments"); return PINN_ERR_INVALID; }
  // one workspace must serve every call on this network: gradient calls that AUTO routes to another engine than the
  // forward (k = 1 networks: fused forward, generic gradient), and the calls that run the network WITHOUT its tangents
  // (pinn_forward, pinn_mse_loss_grad, pinn_jet_backward without gdY: k = 0), which may land on yet another engine
  // (k = 1 at width 65..256: jets on the generic kernels, plain forwards on the wide engine)
  auto need = [&](const Net& nn, int* err) -> int64_t {
    auto ws_of = [&](int e) {
      return e == PINN_ENGINE_FUSED ? fused_workspace_bytes(nn, N)
           : e == PINN_ENGINE_WIDE ? wide_workspace_bytes(nn, N) : generic_workspace_bytes(nn, N);
    };
    int rc1 = PINN_OK, rc2 = PINN_OK;
    const int e = pick_engine(desc, nn, false, &rc1);
    const int eg = pick_engine(desc, nn, true, &rc2);
    if (rc1 && rc2) { *err = rc1; return -1; }      // neither kind of call is served on the engine asked for
    int64_t b = rc1 ? -1 : ws_of(e);
    if (rc2 == PINN_OK && (rc1 || eg != e)) { const int64_t bg = ws_of(eg); if (bg > b) b = bg; }
    return b;
  };
  int64_t b = need(n, &rc);
  if (b < 0 && rc) return rc;
  if (n.k > 0) {
    Net n0 = n; n0.k = 0; n0.K1 = 1;
    int rc0 = PINN_OK;
    const int64_t b0 = need(n0, &rc0);               // (refused on the engine asked for: those calls fail by themselves)
    if (b0 > b) b = b0;
  }
  if (b < 0) { set_error("network not supported"); return PINN_ERR_UNSUPPORTED; }
  *bytes = b;
  return PINN_OK;
}

static int32_t forward_impl(const pinn_desc* desc, const float* params, const float* X, int64_t N, float* Y,
                            float* dY, void* ws, int64_t ws_bytes, void* stream, bool jet) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  if (!params || (!X && N > 0) || N < 0 || (!Y && N > 0) || (jet && !dY && N > 0)) { set_error("NULL pointer argument"); return PINN_ERR_INVALID; }
  if (N == 0) return PINN_OK;
  if (!jet) { n.k = 0; n.K1 = 1; dY = nullptr; }
  if (jet && n.k == 0) { set_error("forward_jet needs k >= 1"); return PINN_ERR_INVALID; }
  const int e = pick_engine(desc, n, false, &rc); if (rc) return rc;
  return e == PINN_ENGINE_FUSED ? fused_forward(n, params, X, N, Y, dY, ws, ws_bytes, (hipStream_t)stream)
       : e == PINN_ENGINE_WIDE ? wide_forward(n, params, X, N, Y, dY, ws, ws_bytes, (hipStream_t)stream)
                               : generic_forward(n, params, X, N, Y, dY, ws, ws_bytes, (hipStream_t)stream);
}

int32_t pinn_forward(const pinn_desc* desc, const float* params, const float* X, int64_t N, float* Y, void* ws,
                     int64_t ws_bytes, void* stream) {
  return forward_impl(desc, params, X, N, Y, nullptr, ws, ws_bytes, stream, false);
}

int32_t pinn_forward_jet(const pinn_desc* desc, const float* params, const float* X, int64_t N, float* Y,
                         float* dY, void* ws, int64_t ws_bytes, void* stream) {
  return forward_impl(desc, params, X, N, Y, dY, ws, ws_bytes, stream, true);
}

int32_t pinn_jet_backward(const pinn_desc* desc, const float* params, const float* X, int64_t N, const float* gY,
                          const float* gdY, float* grad_flat, void* ws, int64_t ws_bytes, void* stream) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  if (!params || (!X && N > 0) || N < 0 || !grad_flat || (!gY && !gdY && N > 0)) { set_error("NULL pointer argument"); return PINN_ERR_INVALID; }
  if (N == 0) return PINN_OK;
  if (!gdY) { n.k = 0; n.K1 = 1; }
  // generic consumer of the jet: the layer-wise engine handles every shape
  return generic_jet_backward(n, params, X, N, gY, gdY, grad_flat, ws, ws_bytes, (hipStream_t)stream);
}

static int32_t residual_impl(const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale,
                             const float* params, const float* X, int64_t N, float* term_sums, float* grad_flat,
                             void* ws, int64_t ws_bytes, void* stream, bool want_grad) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  pinn_residual_spec nspec; rc = check_spec(n, spec, &nspec); if (rc) return rc;
  if (!params || (!X && N > 0) || N < 0 || !term_sums || (want_grad && (!grad_flat || !term_scale))) {
    set_error("NULL pointer argument"); return PINN_ERR_INVALID;
  }
  LossReq rq; memset(&rq, 0, sizeof(rq));
  rq.kind = 0; rq.n_split = -1; rq.spec = nspec; rq.scale = term_scale; rq.sums = term_sums;
  rq.grad = want_grad ? grad_flat : nullptr; rq.n_terms = residual_terms(spec->residual_id);
  if (N == 0) { (void)hipMemsetAsync(term_sums, 0, rq.n_terms * sizeof(float), (hipStream_t)stream); return PINN_OK; }
  const int e = pick_engine(desc, n, want_grad, &rc); if (rc) return rc;
  return e == PINN_ENGINE_FUSED ? fused_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream)
       : e == PINN_ENGINE_WIDE ? wide_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream)
                               : generic_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream);
}

int32_t pinn_residual_loss(const pinn_desc* desc, const pinn_residual_spec* spec, const float* params,
                           const float* X, int64_t N, float* term_sums, void* ws, int64_t ws_bytes, void* stream) {
  return residual_impl(desc, spec, nullptr, params, X, N, term_sums, nullptr, ws, ws_bytes, stream, false);
}

int32_t pinn_residual_loss_grad(const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale,
                                const float* params, const float* X, int64_t N, float* term_sums,
                                float* grad_flat, void* ws, int64_t ws_bytes, void* stream) {
  return residual_impl(desc, spec, term_scale, params, X, N, term_sums, grad_flat, ws, ws_bytes, stream, true);
}

int32_t pinn_mse_loss_grad(const pinn_desc* desc, const float* params, const float* X, const float* T, int64_t N,
                           int32_t n_cols, const int32_t* out_col, const float* col_scale, float* col_sums,
                           float* grad_flat, void* ws, int64_t ws_bytes, void* stream) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  if (!params || ((!X || !T) && N > 0) || N < 0 || !out_col || !col_sums || (grad_flat && !col_scale)) {
    set_error("NULL pointer argument"); return PINN_ERR_INVALID;
  }
  if (n_cols < 1 || n_cols > PINN_MAX_ROLES) { set_error("n_cols=%d outside 1..%d", n_cols, PINN_MAX_ROLES); return PINN_ERR_INVALID; }
  LossReq rq; memset(&rq, 0, sizeof(rq));
  rq.kind = 1; rq.n_split = -1; rq.T = T; rq.n_cols = n_cols; rq.mse_scale = col_scale; rq.mse_sums = col_sums; rq.grad = grad_flat;
  for (int j = 0; j < n_cols; ++j) {
    if (out_col[j] < 0 || out_col[j] >= n.d_out) { set_error("out_col[%d]=%d out of range", j, out_col[j]); return PINN_ERR_INVALID; }
    rq.out_col[j] = out_col[j];
  }
  if (N == 0) { (void)hipMemsetAsync(col_sums, 0, n_cols * sizeof(float), (hipStream_t)stream); return PINN_OK; }
  n.k = 0; n.K1 = 1;  // the fidelity term needs no input derivatives
  const int e = pick_engine(desc, n, grad_flat != nullptr, &rc); if (rc) return rc;
  return e == PINN_ENGINE_FUSED ? fused_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream)
       : e == PINN_ENGINE_WIDE ? wide_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream)
                               : generic_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream);
}

static int32_t residual_mse_impl(int64_t n_split, const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale,
                                    const float* T, int32_t n_cols, const int32_t* out_col, const float* col_scale,
                                    const float* params, const float* X, int64_t N, float* term_sums,
                                    float* col_sums, float* grad_flat, void* ws, int64_t ws_bytes, void* stream) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  pinn_residual_spec nspec; rc = check_spec(n, spec, &nspec); if (rc) return rc;
  if (n_split > N) { set_error("n_res=%lld exceeds N=%lld", (long long)n_split, (long long)N); return PINN_ERR_INVALID; }
  if (!params || ((!X || (!T && n_split != N)) && N > 0) || N < 0 || !term_sums || !col_sums || !out_col || !grad_flat || !term_scale ||
      !col_scale) {
    set_error("NULL pointer argument"); return PINN_ERR_INVALID;
  }
  if (n_cols < 1 || n_cols > PINN_MAX_ROLES) { set_error("n_cols=%d outside 1..%d", n_cols, PINN_MAX_ROLES); return PINN_ERR_INVALID; }
  LossReq rq; memset(&rq, 0, sizeof(rq));
  rq.kind = 2; rq.n_split = n_split; rq.spec = nspec; rq.scale = term_scale; rq.sums = term_sums; rq.n_terms = residual_terms(spec->residual_id);
  rq.T = T; rq.n_cols = n_cols; rq.mse_scale = col_scale; rq.mse_sums = col_sums; rq.grad = grad_flat;
  for (int j = 0; j < n_cols; ++j) {
    if (out_col[j] < 0 || out_col[j] >= n.d_out) { set_error("out_col[%d]=%d out of range", j, out_col[j]); return PINN_ERR_INVALID; }
    rq.out_col[j] = out_col[j];
  }
  if (N == 0) {
    (void)hipMemsetAsync(term_sums, 0, rq.n_terms * sizeof(float), (hipStream_t)stream);
    (void)hipMemsetAsync(col_sums, 0, n_cols * sizeof(float), (hipStream_t)stream);
    return PINN_OK;
  }
  if (n_split == N) {   // no fidelity points: the residual-only pass (T may be NULL)
    rq.kind = 0; rq.n_split = -1;
    (void)hipMemsetAsync(col_sums, 0, n_cols * sizeof(float), (hipStream_t)stream);
  }
  const int e = pick_engine(desc, n, true, &rc); if (rc) return rc;
  return e == PINN_ENGINE_FUSED ? fused_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream)
       : e == PINN_ENGINE_WIDE ? wide_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream)
                               : generic_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream);
}

int32_t pinn_residual_mse_loss_grad(const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale,
                                    const float* T, int32_t n_cols, const int32_t* out_col, const float* col_scale,
                                    const float* params, const float* X, int64_t N, float* term_sums,
                                    float* col_sums, float* grad_flat, void* ws, int64_t ws_bytes, void* stream) {
  return residual_mse_impl(-1, desc, spec, term_scale, T, n_cols, out_col, col_scale, params, X, N, term_sums, col_sums,
                           grad_flat, ws, ws_bytes, stream);
}

int32_t pinn_residual_mse_split_loss_grad(const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale,
                                          const float* T, int32_t n_cols, const int32_t* out_col, const float* col_scale,
                                          const float* params, const float* X, int64_t N, int64_t n_res,
                                          float* term_sums, float* col_sums, float* grad_flat, void* ws,
                                          int64_t ws_bytes, void* stream) {
  if (n_res < 0) { set_error("n_res must be >= 0"); return PINN_ERR_INVALID; }
  return residual_mse_impl(n_res, desc, spec, term_scale, T, n_cols, out_col, col_scale, params, X, N, term_sums,
                           col_sums, grad_flat, ws, ws_bytes, stream);
}

int32_t pinn_adam_step(float* params, const float* grad, float* m, float* v, int64_t P, int64_t step, double lr,
                       double beta1, double beta2, double eps, void* stream) {
  if (!params || !grad || !m || !v || P < 0 || step < 1) { set_error("bad arguments"); return PINN_ERR_INVALID; }
  if (P == 0) return PINN_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad, m,
                     v, P, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)(lr / bc1), (float)sqrt(bc2));
  return check_launch("adam");
}

int32_t pinn_loss_grad_adam_step(const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale,
                                 const float* T, int32_t n_cols, const int32_t* out_col, const float* col_scale,
                                 float* params, const float* X, int64_t N, int64_t n_res, float* term_sums,
                                 float* col_sums, float* grad_flat, const pinn_adam_state* adam, void* ws,
                                 int64_t ws_bytes, void* stream) {
  Net n; int rc = make_net(desc, &n); if (rc) return rc;
  if (!spec || !adam) { set_error("NULL pointer argument"); return PINN_ERR_INVALID; }
  pinn_residual_spec nspec; rc = check_spec(n, spec, &nspec); if (rc) return rc;
  if (!params || !X || N < 1 || !term_sums || !term_scale || !grad_flat || !adam->m || !adam->v || adam->step < 1) {
    set_error("bad arguments"); return PINN_ERR_INVALID;
  }
  if (n_res > N) { set_error("n_res=%lld exceeds N=%lld", (long long)n_res, (long long)N); return PINN_ERR_INVALID; }
  if (n_cols < 0 || n_cols > PINN_MAX_ROLES) { set_error("n_cols=%d outside 0..%d", n_cols, PINN_MAX_ROLES); return PINN_ERR_INVALID; }
  if (n_cols == 0 && n_res != N) { set_error("no fidelity columns: n_res must equal N"); return PINN_ERR_INVALID; }
  if (n_cols > 0 && (!out_col || !col_scale || !col_sums || (!T && n_res != N))) { set_error("NULL pointer argument"); return PINN_ERR_INVALID; }
  LossReq rq; memset(&rq, 0, sizeof(rq));
  rq.spec = nspec; rq.scale = term_scale; rq.sums = term_sums; rq.n_terms = residual_terms(spec->residual_id);
  rq.T = T; rq.n_cols = n_cols; rq.mse_scale = col_scale; rq.mse_sums = col_sums; rq.grad = grad_flat;
  for (int j = 0; j < n_cols; ++j) {
    if (out_col[j] < 0 || out_col[j] >= n.d_out) { set_error("out_col[%d]=%d out of range", j, out_col[j]); return PINN_ERR_INVALID; }
    rq.out_col[j] = out_col[j];
  }
  if (n_res == N) { rq.kind = 0; rq.n_split = -1; }      // residual term only
  else { rq.kind = 2; rq.n_split = n_res < 0 ? -1 : n_res; }
  const int e = pick_engine(desc, n, true, &rc); if (rc) return rc;
  if (e != PINN_ENGINE_FUSED || !fused_supports_adam(n, rq, N)) {
    set_error("pinn_loss_grad_adam_step: needs a one-pass request on the fused engine (use the loss call + pinn_adam_step)");
    return PINN_ERR_UNSUPPORTED;
  }
  AdamReq a;
  const double bc1 = 1.0 - pow(adam->beta1, (double)adam->step);
  const double bc2 = 1.0 - pow(adam->beta2, (double)adam->step);
  a.params = params; a.m = adam->m; a.v = adam->v;
  a.w1 = (float)(1.0 - adam->beta1); a.b2 = (float)adam->beta2; a.w2 = (float)(1.0 - adam->beta2); a.eps = (float)adam->eps;
  a.step_size = (float)(adam->lr / bc1); a.bc2_sqrt = (float)sqrt(bc2);
  a.packed_valid = adam->packed_valid != 0;
  if (adam->n_loss_rows < 0 || adam->n_loss_rows > 8 || (adam->n_loss_rows > 0 && (!adam->loss_rows || !adam->losses))) {
    set_error("bad loss_rows arguments"); return PINN_ERR_INVALID;
  }
  a.n_loss_rows = adam->n_loss_rows; a.loss_rows = adam->loss_rows; a.losses = adam->losses;
  rq.adam = &a;
  if (rq.kind == 0 && n_cols > 0) (void)hipMemsetAsync(col_sums, 0, n_cols * sizeof(float), (hipStream_t)stream);
  return fused_loss(n, rq, params, X, N, ws, ws_bytes, (hipStream_t)stream);
}

int32_t pinn_adam_loop(const pinn_desc* desc, const pinn_residual_spec* spec, const float* term_scale, const float* T,
                       int32_t n_cols, const int32_t* out_col, const float* col_scale, float* params, const float* X,
                       int64_t N, int64_t n_res, float* term_sums, float* col_sums, float* grad_flat,
                       const pinn_adam_state* adam, int32_t n_iters, const double* lr, void* ws, int64_t ws_bytes,
                       void* stream) {
  if (!adam || !lr || n_iters < 0) { set_error("bad arguments"); return PINN_ERR_INVALID; }
  for (int32_t i = 0; i < n_iters; ++i) {
    pinn_adam_state st = *adam;
    st.step = adam->step + i; st.lr = lr[i];
    st.packed_valid = (i > 0 || adam->packed_valid) ? 1 : 0;
    if (adam->n_loss_rows > 0 && adam->losses) st.losses = adam->losses + (int64_t)i * adam->n_loss_rows;
    const int32_t rc = pinn_loss_grad_adam_step(desc, spec, term_scale, T, n_cols, out_col, col_scale, params, X, N, n_res,
                                                term_sums, col_sums, grad_flat, &st, ws, ws_bytes, stream);
    if (rc) return rc;      // (unsupported requests are refused by the first iteration, before anything is launched)
  }
  return PINN_OK;
}

}  // extern "C"
