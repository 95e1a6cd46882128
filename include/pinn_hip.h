/*
 * pinn_hip.h — C-ABI of libpinn_hip.so, the MI355X (gfx950) engine for the
 * PINN depth-inversion hot path: tanh-MLP forward + first-order input-Jacobian
 * ("jet") + PDE residual loss + parameter gradient.
 *
 * The reference (rezasalatin/PINN_depthEstimation) has no FFI; its boundary is
 * the Python call surface.  Every entry point below names the reference code it
 * replaces (file:line under /root/reference):
 *
 *   pinn_forward            dnn.py:54-55          DNN.forward (nn.Sequential of Linear/Tanh)
 *   pinn_forward_jet        dnn.py:54-55 + physics.py:6-15   forward plus every
 *                           compute_gradient(out_c, in_j) column in one pass
 *   pinn_jet_backward       the double-backward torch runs under loss.backward()
 *                           (train.py:191) for a generic consumer of the jet
 *   pinn_residual_loss[_grad] physics.py:18-33,37-47,50-88,91-120 (continuity_only,
 *                           continuity_ftemp, Navier_Stokes, physics_equation)
 *                           fused with loss.backward() (train.py:154,191)
 *   pinn_mse_loss_grad      train.py:131-141 (weighted fidelity MSE) + backward
 *   pinn_residual_mse_loss_grad  train_newmethod.py:122-159 (both on one forward) + backward
 *   pinn_residual_mse_split_loss_grad  train.py:131-157 (fidelity set + collocation set, one launch)
 *   pinn_lbfgs_push / pinn_lbfgs_direction  torch.optim.LBFGS's two-loop recursion (train.py:116-125,200)
 *   pinn_adam_step          torch.optim.Adam.step as called at train.py:192
 *   pinn_loss_grad_adam_step  train.py:189-193 (loss_func + backward + Adam.step) in two launches
 *   pinn_adam_loop          train.py:188-193, n iterations of the above enqueued by one call
 *
 * Conventions
 *   - plain C, no exceptions; every function returns 0 on success, <0 on error;
 *     pinn_last_error() returns a thread-local message for the last failure.
 *   - every pointer named params/X/Y/dY/T/grad/... is a DEVICE pointer owned by
 *     the caller; the library allocates nothing that outlives a call.
 *   - params: flat fp32 [W_0, b_0, W_1, b_1, ... W_L, b_L]; W_l is (out_l, in_l)
 *     row-major — torch's nn.Linear.weight layout, state_dict order
 *     layers.layer_{l}.weight / .bias (dnn.py:32-35).
 *   - X: (N, d_in) row-major fp32 (what torch.cat([... (N,1) ...], -1) yields,
 *     train.py:132,148).  Y: (N, d_out) row-major.  dY: (k, N, d_out): dY[j] is
 *     d Y / d X[:, dir_col[j]] per point.
 *   - stream: a hipStream_t passed as void* (torch's current stream).  Calls only
 *     enqueue work; they never synchronise the device or the stream.
 *   - workspace: query with pinn_query_workspace, allocate once, pass to calls.
 */
#ifndef PINN_HIP_H
#define PINN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PINN_ABI_VERSION 3

#define PINN_MAX_DIRS 3   /* tangent directions (inputs with requires_grad) */
#define PINN_MAX_ROLES 8

/* activation (dnn.py:18-21) */
#define PINN_ACT_TANH 0        /* init_type == 'xavier'  */
#define PINN_ACT_LEAKY_RELU 1  /* init_type == 'kaiming', negative_slope 0.01 */

/* engine selector (0 lets the library pick the fastest kernel that supports the shape) */
#define PINN_ENGINE_AUTO 0
#define PINN_ENGINE_GENERIC 1  /* layer-by-layer VALU kernels, any shape */
#define PINN_ENGINE_FUSED 2    /* MFMA chain kernel, one persistent launch, hidden width <= 64 */
#define PINN_ENGINE_WIDE 3     /* 64 < hidden width <= 256.  PINN_PREC_F32: one launch per layer, jets in HBM (wide_kernel.h);
                                * PINN_PREC_BF16: the chain engine, three kernels for all hidden matrices (chain_kernel.h) */
/* sub-values of PINN_ENGINE_FUSED: which of its kernels runs (AUTO / FUSED choose by point count).
 * They are part of the descriptor, not of the process environment: the library reads no
 * environment variables and keeps no mutable state that changes results. */
#define PINN_ENGINE_FUSED_TILE 4  /* one wave per 16-point tile (the large-N kernel) */
#define PINN_ENGINE_FUSED_COOP 5  /* four waves per tile (small point sets; padded hidden width 64 only) */
#define PINN_ENGINE_FUSED_BATCH 6 /* layer-major batches of tiles per wave (narrow nets: hidden width <= 32, tanh, gradient
                                   * passes; AUTO picks it from 4096 points; other requests fall back to _TILE) */

/* GEMM operand precision.  Everything outside the MFMAs (tanh, residual, adjoints, gradient
 * accumulation, Adam) is fp32 in both modes. */
#define PINN_PREC_F32 0   /* v_mfma_f32_16x16x4_f32: exact fp32 (the reference's precision) */
#define PINN_PREC_BF16 1  /* v_mfma_f32_16x16x32_bf16: bf16 operands (weights split hi + lo, jets rounded to bf16), fp32 accumulate
                           * (BASELINE configs[3]); wide engine only.  A tolerance mode, not fp32 parity: loss / gradient within
                           * 5e-3 of the reference at 12 x 256 (measured 2.8e-3 / 3.3e-3, tests/test_config3_gpu.py) */

/* error codes */
#define PINN_OK 0
#define PINN_ERR_INVALID (-1)
#define PINN_ERR_UNSUPPORTED (-2)
#define PINN_ERR_WORKSPACE (-3)
#define PINN_ERR_LAUNCH (-4)

typedef struct pinn_desc {
  int32_t d_in;       /* config layers.input_features  (train.py:52) */
  int32_t d_out;      /* config layers.output_features (train.py:55) */
  int32_t n_hidden;   /* config layers.hidden_layers   (train.py:53) */
  int32_t width;      /* config layers.hidden_width    (train.py:54) */
  int32_t k;          /* number of inputs with requires_grad "true" (train.py:87) */
  int32_t dir_col[PINN_MAX_DIRS]; /* X column of tangent direction j */
  int32_t activation; /* PINN_ACT_* */
  int32_t engine;     /* PINN_ENGINE_* */
  int32_t precision;  /* PINN_PREC_*: operand type of the weight GEMMs */
  /* nn.Dropout(dropout_rate) after every hidden activation (dnn.py:38) while the module is in training mode
   * (train.py:186).  0 = identity (eval mode, and every config the reference ships).  The keep mask of unit f of
   * hidden layer l at point n is a pure function of (dropout_seed, l, f, n) — pinn_dropout_keep below — so a
   * forward call and the reverse sweep that follows it (same seed) see the same mask without storing it; the caller
   * draws a new seed per forward pass.  Kept units are scaled by 1 / (1 - p), tangents included.
   * Gradient passes of tanh networks of hidden width 33..64 run on the fused tile kernel's dropout instances (the mask
   * re-derived in registers); every other call with dropout_p > 0 runs on the generic engine.  AUTO picks accordingly;
   * FUSED is refused for requests it does not serve, WIDE always. */
  float dropout_p;
  uint32_t dropout_seed;
} pinn_desc;

/* residual ids */
#define PINN_RES_NAVIER_STOKES 1     /* physics.py:50-88  roles out: h,z,u,v   dirs: t,x,y */
#define PINN_RES_PHYSICS_EQUATION 2  /* physics.py:91-120 roles out: h,U,V,eta_mean,Hrms,k  dirs: x,y */
#define PINN_RES_CONTINUITY_FTEMP 3  /* physics.py:37-47  roles out: h,U,V     dirs: x,y */
#define PINN_RES_CONTINUITY_ONLY 4   /* physics.py:18-33  same + h anchor where x < 25.5 */

/* number of loss terms each residual reports (sums of squares, un-normalised) */
#define PINN_NS_TERMS 3   /* sum fc^2, sum fm_x^2, sum fm_y^2 */
#define PINN_PE_TERMS 3   /* sum fc^2, sum fx^2,  sum fy^2  */
#define PINN_CF_TERMS 1   /* sum fc^2 */
#define PINN_CO_TERMS 3   /* sum fc^2, sum_{x<thr} (h-anchor)^2, count{x<thr} */

typedef struct pinn_residual_spec {
  int32_t residual_id;
  int32_t out_col[PINN_MAX_ROLES]; /* output column of each role, in the role order above; the network may have MORE
                                      output columns than roles, in any order (entries beyond the residual's roles
                                      are ignored, whatever they hold) */
  int32_t dir_of[PINN_MAX_DIRS];   /* index into desc.dir_col of each direction role (entries beyond the residual's
                                      directions are ignored) */
  int32_t flags;                   /* bit0: 1 = "corrected" radiation stress (unused; E==0 bug-compatible, physics.py:106) */
  float param[4];                  /* continuity_only: param[0]=threshold (25.5), param[1]=anchor (0.75) */
} pinn_residual_spec;

int32_t pinn_version(void);

/* 1 if unit `feature` of hidden layer `layer` (0-based) is kept at point `point` under (seed, p) — the exact mask
 * the kernels apply (host evaluation of the same function; for tests and for callers that want the mask). */
int32_t pinn_dropout_keep(uint32_t seed, int32_t layer, int32_t feature, int64_t point, float p);
const char* pinn_last_error(void);

/* P = sum_l (in_l*out_l + out_l), layers = [d_in] + [width]*n_hidden + [d_out] (train.py:56) */
int32_t pinn_param_count(const pinn_desc* desc, int64_t* count);

/* bytes of workspace a call on N points needs with the engine desc->engine selects.
 * pinn_jet_backward always runs on the generic engine: query with
 * engine = PINN_ENGINE_GENERIC for it. */
int32_t pinn_query_workspace(const pinn_desc* desc, int64_t N, int64_t* bytes);

int32_t pinn_forward(const pinn_desc* desc, const float* params, const float* X, int64_t N,
                     float* Y, void* ws, int64_t ws_bytes, void* stream);

int32_t pinn_forward_jet(const pinn_desc* desc, const float* params, const float* X, int64_t N,
                         float* Y, float* dY, void* ws, int64_t ws_bytes, void* stream);

/* grad_flat (P,) += d/dtheta [ sum(gY*Y) + sum(gdY*dY) ];  gdY may be NULL (treated as 0) */
int32_t pinn_jet_backward(const pinn_desc* desc, const float* params, const float* X, int64_t N,
                          const float* gY, const float* gdY, float* grad_flat,
                          void* ws, int64_t ws_bytes, void* stream);

/* term_sums[t] = sum over points of (residual field t)^2  (device, n_terms floats, overwritten) */
int32_t pinn_residual_loss(const pinn_desc* desc, const pinn_residual_spec* spec,
                           const float* params, const float* X, int64_t N,
                           float* term_sums, void* ws, int64_t ws_bytes, void* stream);

/* as above, and grad_flat (P,) += sum_t term_scale[t] * d term_sums[t] / d theta.
 * term_scale is a DEVICE array (n_terms floats) so that a data-dependent
 * normaliser (continuity_only's count, a global N under data parallelism) never
 * needs a host round trip.  For mean-of-squares losses term_scale[t] = weight/N. */
int32_t pinn_residual_loss_grad(const pinn_desc* desc, const pinn_residual_spec* spec,
                                const float* term_scale,
                                const float* params, const float* X, int64_t N,
                                float* term_sums, float* grad_flat,
                                void* ws, int64_t ws_bytes, void* stream);

/* fidelity: col_sums[j] = sum_n (T[n,j] - Y[n,out_col[j]])^2 ;
 * grad_flat += sum_j col_scale[j] * d col_sums[j] / d theta   (train.py:136-141).
 * T is (N, n_cols) row-major; out_col is a HOST array; col_scale/col_sums are device. */
int32_t pinn_mse_loss_grad(const pinn_desc* desc, const float* params, const float* X,
                           const float* T, int64_t N, int32_t n_cols, const int32_t* out_col,
                           const float* col_scale, float* col_sums, float* grad_flat,
                           void* ws, int64_t ws_bytes, void* stream);

/* residual + fidelity on ONE point set in one pass (train_newmethod.py:122-159: a single forward
 * feeds both F.mse_loss on the `trues` columns and the residual):
 * grad_flat += sum_t term_scale[t] d term_sums[t]/d theta + sum_j col_scale[j] d col_sums[j]/d theta */
int32_t pinn_residual_mse_loss_grad(const pinn_desc* desc, const pinn_residual_spec* spec,
                                    const float* term_scale, const float* T, int32_t n_cols,
                                    const int32_t* out_col, const float* col_scale,
                                    const float* params, const float* X, int64_t N,
                                    float* term_sums, float* col_sums, float* grad_flat,
                                    void* ws, int64_t ws_bytes, void* stream);

/* train.py:131-157 in ONE launch: the reference's loss_func runs the network twice per iteration, on
 * the fidelity points (train.py:132-141) and on the collocation points (train.py:148-154).  Here X
 * holds the n_res collocation points FIRST and the N - n_res fidelity points after them; T holds
 * the fidelity targets only, (N - n_res, n_cols).  The residual terms are summed over the first
 * n_res points, the squared errors over the rest (fidelity points ride through the jet kernels
 * with unused tangents: meant for N_fid << N_res or small N, where launches dominate). */
int32_t pinn_residual_mse_split_loss_grad(const pinn_desc* desc, const pinn_residual_spec* spec,
                                          const float* term_scale, const float* T, int32_t n_cols,
                                          const int32_t* out_col, const float* col_scale,
                                          const float* params, const float* X, int64_t N, int64_t n_res,
                                          float* term_sums, float* col_sums, float* grad_flat,
                                          void* ws, int64_t ws_bytes, void* stream);

/* train.py:189-193 as TWO launches (fused engine, one-pass requests): loss_func's forward + residual + backward, then
 * ONE kernel that finishes the loss sums and the gradient, applies torch.optim.Adam's update (pinn_adam_step's
 * arithmetic, bit for bit) to params / m / v and refreshes the packed weights in `ws` that the next call's first
 * kernel reads.  At the reference's own problem sizes (N_res = 243, config_CMB.json:43) the iteration is bound by
 * its launches: pack + pass + two reductions + zero-fill + update were six of them.
 *   X, n_res: as pinn_residual_mse_split_loss_grad (n_res == N: residual term only, n_cols may be 0;
 *             n_res < 0: both terms on every point, train_newmethod.py:122-159).
 *   grad_flat is OVERWRITTEN with this iteration's gradient (no zero-fill needed).
 *   adam->loss_rows / losses: optional — the finishing kernel also forms the weighted loss values the loop logs
 *             (double accumulation over the few sums), saving the caller a launch per iteration.
 *   adam->packed_valid: nonzero iff the previous call on this `ws` was this function with the same desc and
 *             nothing has written params since — the call then skips the packing kernel.
 * Returns PINN_ERR_UNSUPPORTED, having launched nothing, when the request would not run as one pass of the fused
 * engine (wide / generic engines, large split requests): use the loss call followed by pinn_adam_step. */
typedef struct pinn_adam_state {
  float* m;              /* exp_avg    (P) */
  float* v;              /* exp_avg_sq (P) */
  int64_t step;          /* 1-based, as torch counts */
  double lr, beta1, beta2, eps;
  int32_t packed_valid;
  int32_t n_loss_rows;   /* 0, or: also write losses[r] = sum_j loss_rows[r][j] * S_j, S = [col_sums (n_cols) | term_sums] */
  const float* loss_rows;/* (n_loss_rows, n_cols + n_terms) row-major weights (train.py:141,154,157: fidelity / residual / total) */
  float* losses;         /* (n_loss_rows) */
} pinn_adam_state;
int32_t pinn_loss_grad_adam_step(const pinn_desc* desc, const pinn_residual_spec* spec,
                                 const float* term_scale, const float* T, int32_t n_cols,
                                 const int32_t* out_col, const float* col_scale,
                                 float* params, const float* X, int64_t N, int64_t n_res,
                                 float* term_sums, float* col_sums, float* grad_flat,
                                 const pinn_adam_state* adam, void* ws, int64_t ws_bytes, void* stream);

/* n_iters consecutive Adam iterations of pinn_loss_grad_adam_step on the same point set (train.py:188-193, the
 * `for epoch in range(adam_maxit)` loop without the host in it: 2 n_iters launches enqueued by one call; at the
 * reference's problem sizes the Python-side cost of an iteration had become as large as its kernels).  Iteration i uses
 * step adam->step + i and learning rate lr[i] (HOST array: the StepLR schedule, train.py:109-113); when
 * adam->n_loss_rows > 0 it writes its weighted losses to adam->losses + i * n_loss_rows.  adam->lr is ignored.
 * Same refusal rule as pinn_loss_grad_adam_step (nothing is launched when unsupported). */
int32_t pinn_adam_loop(const pinn_desc* desc, const pinn_residual_spec* spec,
                       const float* term_scale, const float* T, int32_t n_cols,
                       const int32_t* out_col, const float* col_scale,
                       float* params, const float* X, int64_t N, int64_t n_res,
                       float* term_sums, float* col_sums, float* grad_flat,
                       const pinn_adam_state* adam, int32_t n_iters, const double* lr,
                       void* ws, int64_t ws_bytes, void* stream);

/* torch.optim.Adam single-tensor update on flat buffers (amsgrad off, weight_decay 0,
 * maximize off): m,v are exp_avg / exp_avg_sq; step is the 1-based step count;
 * lr is a host double (StepLR changes it between steps, train.py:193); the scalar
 * factors are formed in double as Python forms them and cast to fp32 once. */
int32_t pinn_adam_step(float* params, const float* grad, float* m, float* v, int64_t P,
                       int64_t step, double lr, double beta1, double beta2, double eps,
                       void* stream);

/* L-BFGS two-loop recursion of torch.optim.LBFGS (train.py:116-125, one .step(closure), train.py:200)
 * on device.  S, Y: (m x P) row-major rings of steps and gradient differences, M = S Y^T (m x m, fp64,
 * physical row indices); logical pair i (0 = oldest) is physical row (head + i) % m; k pairs in use.
 * pinn_lbfgs_push stores (s, y) in row `slot` and refreshes row and column `slot` of M.
 * pinn_lbfgs_direction writes d = -H_k g (H the initial scaling ys/yy); tmp: 4m doubles, coef: 2m
 * floats, q: P floats of scratch.  Unused rows of S, Y must be zero.  m <= 256. */
int32_t pinn_lbfgs_push(float* S, float* Y, double* M, int32_t m, int64_t P, int32_t slot,
                        const float* s, const float* y, void* stream);
int32_t pinn_lbfgs_direction(const float* S, const float* Y, const double* M, int32_t m, int64_t P,
                             int32_t head, int32_t k, const float* g, double H, float* d,
                             double* tmp, float* coef, float* q, void* stream);

/* ---- staging of the collocation points on the device (train.py:246-277, operations.py:4-30) --------------------
 * The reference loads each input variable as a (ny, nx) float64 grid (scipy.io.loadmat), subsamples it with
 * [::interval_x, ::interval_y] (train.py:260), maps it onto [-1, 1] with the variable's (min, max) (operations.py:4-8;
 * a degenerate range gives zeros), flattens it COLUMN-major (train.py:265-267), stacks the variables as columns and
 * drops every row that holds a NaN (train.py:276-277).  These two calls do the same on grids already resident on the
 * device, in float64 and in NumPy's order of operations, casting to fp32 last (train.py:88): bit-identical to the
 * host path.  pinn_nanminmax_f64 is np.nanmin / np.nanmax (operations.py:26-27; {NaN, NaN} for an all-NaN array).
 * grids: HOST array of d_in DEVICE pointers (ny * nx doubles each, row-major); minmax: device (d_in, 2) doubles;
 * X_out: device, room for ceil(ny/ix) * ceil(nx/iy) rows of d_in floats; n_rows_out: device, rows actually written. */
int32_t pinn_nanminmax_f64(const double* data, int64_t n, double* out2, void* ws, int64_t ws_bytes, void* stream);
int64_t pinn_stage_workspace_bytes(int64_t ny, int64_t nx, int32_t interval_x, int32_t interval_y);
int32_t pinn_stage_grid_columns(const double* const* grids, int32_t d_in, int64_t ny, int64_t nx, int32_t interval_x,
                                int32_t interval_y, const double* minmax, float* X_out, int64_t* n_rows_out,
                                void* ws, int64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PINN_HIP_H */
