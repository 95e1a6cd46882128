// pinn_generic.hip — shape-agnostic engine: one thread per collocation point,
// one kernel per layer, activations kept in a feature-major workspace
// [quantity c][feature f][point n] so every access is coalesced along n.
//
// It exists so that EVERY network the reference's configs describe (10x10,
// 100x20, 12x256 ...) runs on the GPU, and as an independent on-device check of
// the fused MFMA engine.  HBM-bound by construction (activations round-trip
// through memory each layer); the fused engine is the fast path.
//
// Math: forward-mode jet through each layer
//   z = W a + b, zdot_j = W adot_j ; a' = act(z), adot'_j = act'(z) * zdot_j
// (dnn.py:54-55 + physics.py:6-15), then one reverse sweep for d loss / d theta
// (replaces the double backward of train.py:191):
//   zbar_j' = s * abar_j' ;  zbar = s * abar - 2 a' * sum_j abar_j' * adot'_j   (tanh, s = 1-a'^2)
//   dW += zbar (x) a + sum_j zbar_j (x) adot_j ;  db += zbar ;  abar = W^T zbar ...
#include <type_traits>
#include "common.h"
#include "residuals.h"

namespace pinn {

namespace {

constexpr int TPB = 256;
constexpr int OB = 8;  // outputs register-blocked per pass

__device__ inline float act_fwd(int act, float z) {
  if (act == PINN_ACT_TANH) return tanh_f32(z);
  return z > 0.f ? z : 0.01f * z;  // nn.LeakyReLU(0.01) dnn.py:21
}
__device__ inline float act_slope(int act, float a) {  // act'(z) expressed through a = act(z)
  if (act == PINN_ACT_TANH) return 1.f - a * a;
  return a > 0.f ? 1.f : 0.01f;
}

// a0[c][i][n]: c = 0 -> X[n][i]; c = 1+j -> unit tangent e_{dir_col[j]}
template <int K1>
__global__ void k_seed(const float* __restrict__ X, int d_in, int64_t N, int dir0, int dir1, int dir2,
                       float* __restrict__ a0) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int dirs[3] = {dir0, dir1, dir2};
  for (int i = 0; i < d_in; ++i) {
    a0[(int64_t)i * N + n] = X[n * d_in + i];
#pragma unroll
    for (int c = 1; c < K1; ++c) a0[((int64_t)c * d_in + i) * N + n] = (dirs[c - 1] == i) ? 1.f : 0.f;
  }
}

// nn.Dropout(p) in training mode (dnn.py:38): what the kernels need of it
struct Drop {
  uint32_t thresh;   // 0 = off
  uint32_t seed;
  int layer;
  float scale, keep_p;   // 1 / (1 - p), 1 - p
};

// mode: 0 = linear output layer, 1 = hidden layer (activation applied, then dropout: a <- m a / (1-p), the
// tangents likewise — the same mask multiplies value and derivatives, as autograd through nn.Dropout does)
template <int K1>
__global__ void k_fwd_layer(const float* __restrict__ Wt, const float* __restrict__ b, int in_dim, int out_dim,
                            const float* __restrict__ a_in, float* __restrict__ a_out, int64_t N, int hidden,
                            int act, Drop dr) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int o0 = 0; o0 < out_dim; o0 += OB) {
    float acc[K1][OB];
#pragma unroll
    for (int j = 0; j < OB; ++j) {
      acc[0][j] = (o0 + j < out_dim) ? b[o0 + j] : 0.f;
#pragma unroll
      for (int c = 1; c < K1; ++c) acc[c][j] = 0.f;
    }
    for (int i = 0; i < in_dim; ++i) {
      float av[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) av[c] = a_in[((int64_t)c * in_dim + i) * N + n];
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const float w = (o0 + j < out_dim) ? Wt[(int64_t)(o0 + j) * in_dim + i] : 0.f;
#pragma unroll
        for (int c = 0; c < K1; ++c) acc[c][j] = fmaf(w, av[c], acc[c][j]);
      }
    }
#pragma unroll
    for (int j = 0; j < OB; ++j) {
      if (o0 + j >= out_dim) break;
      float a = acc[0][j], s = 1.f;
      if (hidden) {
        a = act_fwd(act, a); s = act_slope(act, a);
        if (dr.thresh) {
          const float m = dropout_bits(dr.seed, dr.layer, o0 + j, n) >= dr.thresh ? dr.scale : 0.f;
          a *= m; s *= m;
        }
      }
      a_out[((int64_t)(o0 + j)) * N + n] = a;
#pragma unroll
      for (int c = 1; c < K1; ++c) a_out[((int64_t)c * out_dim + o0 + j) * N + n] = s * acc[c][j];
    }
  }
}

// g (in/out): on entry abar' (adjoint of this layer's OUTPUT jet), on exit zbar.
// g_in: adjoint of the layer's input jet (skipped when need_gin == 0).
// With dropout the stored jet is masked and scaled: a_out = m c t (t the activation value, c = 1 / (1 - p)), its
// tangents m c t' zdot.  Then  d a_out / dz = m c t',  d adot_out / dz = -2 t adot_out (tanh)  and
// d adot_out / dzdot = m c t': the formulas below with  a := t = a_out (1 - p)  and  s := m c t'  (the mask is
// re-derived from the forward's seed; for a dropped unit a_out = adot_out = 0 and s = 0, so zbar = 0).
template <int K1>
__global__ void k_bwd_layer(const float* __restrict__ Wt, int in_dim, int out_dim, float* __restrict__ g,
                            const float* __restrict__ a_out, float* __restrict__ g_in, int64_t N, int hidden,
                            int act, int need_gin, Drop dr) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (hidden) {
    for (int o = 0; o < out_dim; ++o) {
      float a = a_out[(int64_t)o * N + n];
      float s;
      if (dr.thresh) {
        a *= dr.keep_p;
        s = dropout_bits(dr.seed, dr.layer, o, n) >= dr.thresh ? dr.scale * act_slope(act, a) : 0.f;
      } else s = act_slope(act, a);
      float gb = g[(int64_t)o * N + n];
      float cross = 0.f;
#pragma unroll
      for (int c = 1; c < K1; ++c) {
        const int64_t idx = ((int64_t)c * out_dim + o) * N + n;
        const float gd = g[idx];
        cross = fmaf(gd, a_out[idx], cross);
        g[idx] = s * gd;
      }
      float zb = s * gb;
      if (act == PINN_ACT_TANH) zb = fmaf(-2.f * a, cross, zb);  // tanh'' = -2 a (1-a^2)
      g[(int64_t)o * N + n] = zb;
    }
  }
  if (!need_gin) return;
  for (int i0 = 0; i0 < in_dim; i0 += OB) {
    float acc[K1][OB];
#pragma unroll
    for (int c = 0; c < K1; ++c)
#pragma unroll
      for (int j = 0; j < OB; ++j) acc[c][j] = 0.f;
    for (int o = 0; o < out_dim; ++o) {
      float zb[K1];
#pragma unroll
      for (int c = 0; c < K1; ++c) zb[c] = g[((int64_t)c * out_dim + o) * N + n];
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const float w = (i0 + j < in_dim) ? Wt[(int64_t)o * in_dim + i0 + j] : 0.f;
#pragma unroll
        for (int c = 0; c < K1; ++c) acc[c][j] = fmaf(w, zb[c], acc[c][j]);
      }
    }
#pragma unroll
    for (int j = 0; j < OB; ++j) {
      if (i0 + j >= in_dim) break;
#pragma unroll
      for (int c = 0; c < K1; ++c) g_in[((int64_t)c * in_dim + i0 + j) * N + n] = acc[c][j];
    }
  }
}

// dW[o][i] += sum_n sum_c zbar[c][o][n] * a_in[c][i][n] ; db[o] += sum_n zbar[0][o][n]
// block = 16x16 threads owning a 16x16 tile of dW; grid.z walks point chunks.
constexpr int WG_PTS = 64;
template <int K1>
__global__ void k_wgrad(const float* __restrict__ zb, const float* __restrict__ a_in, int in_dim, int out_dim,
                        int64_t N, int64_t chunk, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float zs[K1][16][WG_PTS + 1];
  __shared__ float as[K1][16][WG_PTS + 1];
  const int ti = threadIdx.x & 15, to = threadIdx.x >> 4;
  const int o0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
  const int64_t n_begin = (int64_t)blockIdx.z * chunk;
  const int64_t n_end = (n_begin + chunk < N) ? n_begin + chunk : N;
  float acc = 0.f, accb = 0.f;
  for (int64_t nb = n_begin; nb < n_end; nb += WG_PTS) {
    for (int e = threadIdx.x; e < K1 * 16 * WG_PTS; e += 256) {
      const int p = e % WG_PTS, f = (e / WG_PTS) % 16, c = e / (WG_PTS * 16);
      const int64_t n = nb + p;
      const bool ok = n < n_end;
      zs[c][f][p] = (ok && o0 + f < out_dim) ? zb[((int64_t)c * out_dim + o0 + f) * N + n] : 0.f;
      as[c][f][p] = (ok && i0 + f < in_dim) ? a_in[((int64_t)c * in_dim + i0 + f) * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < K1; ++c)
      for (int p = 0; p < WG_PTS; ++p) acc = fmaf(zs[c][to][p], as[c][ti][p], acc);
    if (blockIdx.x == 0 && ti == 0)
      for (int p = 0; p < WG_PTS; ++p) accb += zs[0][to][p];
    __syncthreads();
  }
  if (o0 + to < out_dim && i0 + ti < in_dim) atomicAdd(&dW[(int64_t)(o0 + to) * in_dim + i0 + ti], acc);
  if (blockIdx.x == 0 && ti == 0 && o0 + to < out_dim) atomicAdd(&db[o0 + to], accb);
}

// ---- loss kernels ------------------------------------------------------------------
template <int NT>
__device__ inline void block_reduce_store(float (&t)[NT], float* __restrict__ partial) {
  __shared__ float red[NT][TPB / 64];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    float v = t[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < NT) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) v += red[threadIdx.x][w];
    partial[(int64_t)blockIdx.x * NT + threadIdx.x] = v;
  }
}

struct RoleMap {
  int out_col[PINN_MAX_ROLES];
  int q_of[PINN_MAX_DIRS];  // quantity index (1 + direction index) of each direction role
};

template <class RES, bool GRAD>
__global__ void k_residual(const float* __restrict__ yj, int d_out, int64_t N, RoleMap rm,
                           const float* __restrict__ scale, const float* __restrict__ X, int d_in, int xcol,
                           int anchor_on, float thr, float anchor, float* __restrict__ G,
                           float* __restrict__ partial, int64_t n_res) {
  constexpr int NR = RES::NR, ND = RES::ND, NT = RES::NT;
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float sq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) sq[t] = 0.f;
  if (n < n_res) {   // (split mode: only the first n_res points are collocation points)
    float v[1 + ND][NR], g[1 + ND][NR];
#pragma unroll
    for (int c = 0; c <= ND; ++c) {
      const int q = (c == 0) ? 0 : rm.q_of[c - 1];
#pragma unroll
      for (int r = 0; r < NR; ++r) v[c][r] = yj[((int64_t)q * d_out + rm.out_col[r]) * N + n];
    }
    float sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sc[t] = GRAD ? scale[t] : 0.f;
    if constexpr (std::is_same<RES, ResContinuity>::value) {
      const bool masked = anchor_on ? (X[n * d_in + xcol] < thr) : false;
      RES::template eval<GRAD>(v, sc, g, sq, anchor_on != 0, masked, anchor);
    } else {
      RES::template eval<GRAD>(v, sc, g, sq);
    }
    if (GRAD) {
#pragma unroll
      for (int c = 0; c <= ND; ++c) {
        const int q = (c == 0) ? 0 : rm.q_of[c - 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) G[((int64_t)q * d_out + rm.out_col[r]) * N + n] = g[c][r];
      }
    }
  }
  block_reduce_store<NT>(sq, partial);
}

// fidelity: (true - pred)^2 per column (train.py:141)
struct MseMap { int n_cols; int out_col[PINN_MAX_ROLES]; };
template <bool GRAD>
__global__ void k_mse(const float* __restrict__ y, int d_out, int64_t N, MseMap mm, const float* __restrict__ T,
                      const float* __restrict__ scale, float* __restrict__ G, float* __restrict__ partial, int64_t n0) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float sq[PINN_MAX_ROLES];
#pragma unroll
  for (int j = 0; j < PINN_MAX_ROLES; ++j) sq[j] = 0.f;
  if (n >= n0 && n < N) {   // (split mode: fidelity points start at n0; T holds their rows only)
#pragma unroll
    for (int j = 0; j < PINN_MAX_ROLES; ++j) {
      if (j < mm.n_cols) {
        const int64_t idx = (int64_t)mm.out_col[j] * N + n;
        const float d = T[(n - n0) * mm.n_cols + j] - y[idx];
        sq[j] = d * d;
        if (GRAD) G[idx] += -2.f * scale[j] * d;  // G pre-zeroed; += lets two targets share a column
      }
    }
  }
  block_reduce_store<PINN_MAX_ROLES>(sq, partial);
}

__global__ void k_reduce_partials(const float* __restrict__ partial, int64_t nblocks, int nt_stride, int nt,
                                  float* __restrict__ out) {
  // one block per term; fixed summation order -> deterministic
  const int t = blockIdx.x;
  if (t >= nt) return;
  __shared__ double red[256];
  double v = 0.0;
  for (int64_t b = threadIdx.x; b < nblocks; b += 256) v += (double)partial[b * nt_stride + t];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[t] = (float)red[0];
}

// G[c][o][n] from row-major gY (N,d_out) / gdY (k,N,d_out)
template <int K1>
__global__ void k_seed_adjoint(const float* __restrict__ gY, const float* __restrict__ gdY, int d_out, int64_t N,
                               float* __restrict__ G) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int o = 0; o < d_out; ++o) {
    G[(int64_t)o * N + n] = gY ? gY[n * d_out + o] : 0.f;
#pragma unroll
    for (int c = 1; c < K1; ++c)
      G[((int64_t)c * d_out + o) * N + n] = gdY ? gdY[((int64_t)(c - 1) * N + n) * d_out + o] : 0.f;
  }
}

template <int K1>
__global__ void k_unseed(const float* __restrict__ out, int d_out, int64_t N, float* __restrict__ Y,
                         float* __restrict__ dY) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int o = 0; o < d_out; ++o) {
    if (Y) Y[n * d_out + o] = out[(int64_t)o * N + n];
    if (dY) {
#pragma unroll
      for (int c = 1; c < K1; ++c) dY[((int64_t)(c - 1) * N + n) * d_out + o] = out[((int64_t)c * d_out + o) * N + n];
    }
  }
}

// ---- workspace carve ---------------------------------------------------------------
inline int maxdim(const Net& n) {
  int m = n.W;
  if (n.d_in > m) m = n.d_in;
  if (n.d_out > m) m = n.d_out;
  return m;
}
inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

struct Layout {
  int64_t act_off[1024 + 2];  // a_0 .. a_L, then out
  int64_t g0, g1, partial, total;
  int64_t nblocks;
};

bool make_layout(const Net& n, int64_t N, Layout* lo) {
  if (n.L + 2 > 1024 + 2) return false;
  const int K1 = 1 + PINN_MAX_DIRS;  // size for the worst case so one workspace serves every call
  int64_t off = 0;
  for (int l = 0; l <= n.L; ++l) {
    lo->act_off[l] = off;
    off += align256((int64_t)K1 * (l == 0 ? n.d_in : n.W) * N * 4);
  }
  lo->act_off[n.L + 1] = off;
  off += align256((int64_t)K1 * n.d_out * N * 4);
  lo->g0 = off; off += align256((int64_t)K1 * maxdim(n) * N * 4);
  lo->g1 = off; off += align256((int64_t)K1 * maxdim(n) * N * 4);
  lo->nblocks = (N + TPB - 1) / TPB;
  lo->partial = off; off += align256(lo->nblocks * PINN_MAX_ROLES * 4);
  lo->total = off;
  return true;
}

template <int K1>
int run_forward(const Net& n, const float* params, const float* X, int64_t N, char* ws, const Layout& lo,
                hipStream_t s) {
  const unsigned grid = (unsigned)lo.nblocks;
  hipLaunchKernelGGL(k_seed<K1>, dim3(grid), dim3(TPB), 0, s, X, n.d_in, N, n.dir_col[0], n.dir_col[1],
                     n.dir_col[2], (float*)(ws + lo.act_off[0]));
  for (int l = 0; l <= n.L; ++l) {
    const Drop dr{n.drop_p > 0.f ? n.drop_thresh : 0u, n.drop_seed, l, 1.f / (1.f - n.drop_p), 1.f - n.drop_p};
    hipLaunchKernelGGL(k_fwd_layer<K1>, dim3(grid), dim3(TPB), 0, s, params + n.w_off(l), params + n.b_off(l),
                       n.in_dim(l), n.out_dim(l), (const float*)(ws + lo.act_off[l]),
                       (float*)(ws + lo.act_off[l + 1]), N, l < n.L ? 1 : 0, n.act, dr);
  }
  return check_launch("generic forward");
}

template <int K1>
int run_backward(const Net& n, const float* params, int64_t N, char* ws, const Layout& lo, float* grad,
                 hipStream_t s) {
  const unsigned grid = (unsigned)lo.nblocks;
  float* gcur = (float*)(ws + lo.g0);
  float* gnext = (float*)(ws + lo.g1);
  int64_t chunk = 16384;
  for (int l = n.L; l >= 0; --l) {
    const int in_dim = n.in_dim(l), out_dim = n.out_dim(l);
    const Drop dr{n.drop_p > 0.f ? n.drop_thresh : 0u, n.drop_seed, l, 1.f / (1.f - n.drop_p), 1.f - n.drop_p};
    hipLaunchKernelGGL(k_bwd_layer<K1>, dim3(grid), dim3(TPB), 0, s, params + n.w_off(l), in_dim, out_dim, gcur,
                       (const float*)(ws + lo.act_off[l + 1]), gnext, N, l < n.L ? 1 : 0, n.act, l > 0 ? 1 : 0, dr);
    dim3 wg((in_dim + 15) / 16, (out_dim + 15) / 16, (unsigned)((N + chunk - 1) / chunk));
    hipLaunchKernelGGL(k_wgrad<K1>, wg, dim3(256), 0, s, (const float*)gcur, (const float*)(ws + lo.act_off[l]),
                       in_dim, out_dim, N, chunk, grad + n.w_off(l), grad + n.b_off(l));
    float* t = gcur; gcur = gnext; gnext = t;
  }
  return check_launch("generic backward");
}

template <int K1>
int run_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N, char* ws,
             const Layout& lo, hipStream_t s) {
  int rc = run_forward<K1>(n, params, X, N, ws, lo, s);
  if (rc) return rc;
  const unsigned grid = (unsigned)lo.nblocks;
  const float* out = (const float*)(ws + lo.act_off[n.L + 1]);
  float* G = (float*)(ws + lo.g0);
  float* partial = (float*)(ws + lo.partial);
  const bool want_grad = rq.grad != nullptr;
  if (want_grad) (void)hipMemsetAsync(G, 0, (size_t)K1 * n.d_out * N * 4, s);
  const int64_t n_res = (rq.kind == 2 && rq.n_split >= 0) ? rq.n_split : N;
  const int64_t n0 = (rq.kind == 2 && rq.n_split >= 0) ? rq.n_split : 0;
  if (rq.kind == 0 || rq.kind == 2) {
    int nt_stride = 0;
    RoleMap rm;
    for (int r = 0; r < PINN_MAX_ROLES; ++r) rm.out_col[r] = rq.spec.out_col[r];
    for (int d = 0; d < PINN_MAX_DIRS; ++d) rm.q_of[d] = 1 + rq.spec.dir_of[d];
    const int id = rq.spec.residual_id;
#define LAUNCH_RES(RES, ANCH, XCOL)                                                                            \
  do {                                                                                                         \
    nt_stride = RES::NT;                                                                                       \
    if (want_grad)                                                                                             \
      hipLaunchKernelGGL((k_residual<RES, true>), dim3(grid), dim3(TPB), 0, s, out, n.d_out, N, rm, rq.scale, X, \
                         n.d_in, XCOL, ANCH, rq.spec.param[0], rq.spec.param[1], G, partial, n_res);           \
    else                                                                                                       \
      hipLaunchKernelGGL((k_residual<RES, false>), dim3(grid), dim3(TPB), 0, s, out, n.d_out, N, rm, rq.scale, X, \
                         n.d_in, XCOL, ANCH, rq.spec.param[0], rq.spec.param[1], G, partial, n_res);           \
  } while (0)
    if (id == PINN_RES_NAVIER_STOKES) LAUNCH_RES(ResNavierStokes, 0, 0);
    else if (id == PINN_RES_PHYSICS_EQUATION) LAUNCH_RES(ResPhysicsEquation, 0, 0);
    else if (id == PINN_RES_CONTINUITY_FTEMP) LAUNCH_RES(ResContinuity, 0, 0);
    else if (id == PINN_RES_CONTINUITY_ONLY) LAUNCH_RES(ResContinuity, 1, n.dir_col[rq.spec.dir_of[0]]);
    else { set_error("unknown residual_id %d", id); return PINN_ERR_INVALID; }
#undef LAUNCH_RES
    hipLaunchKernelGGL(k_reduce_partials, dim3(rq.n_terms), dim3(256), 0, s, (const float*)partial, lo.nblocks,
                       nt_stride, rq.n_terms, rq.sums);
  }
  if (rq.kind == 1 || rq.kind == 2) {   // fidelity columns; adds into G (pre-zeroed / after the residual's writes)
    MseMap mm; mm.n_cols = rq.n_cols;
    for (int j = 0; j < PINN_MAX_ROLES; ++j) mm.out_col[j] = j < rq.n_cols ? rq.out_col[j] : 0;
    if (want_grad) hipLaunchKernelGGL(k_mse<true>, dim3(grid), dim3(TPB), 0, s, out, n.d_out, N, mm, rq.T, rq.mse_scale, G, partial, n0);
    else hipLaunchKernelGGL(k_mse<false>, dim3(grid), dim3(TPB), 0, s, out, n.d_out, N, mm, rq.T, rq.mse_scale, G, partial, n0);
    hipLaunchKernelGGL(k_reduce_partials, dim3(rq.n_cols), dim3(256), 0, s, (const float*)partial, lo.nblocks,
                       PINN_MAX_ROLES, rq.n_cols, rq.mse_sums);
  }
  rc = check_launch("generic loss");
  if (rc || !want_grad) return rc;
  return run_backward<K1>(n, params, N, ws, lo, rq.grad, s);
}

}  // namespace

int64_t generic_workspace_bytes(const Net& n, int64_t N) {
  Layout lo;
  if (!make_layout(n, N > 0 ? N : 1, &lo)) return -1;
  return lo.total;
}

#define DISPATCH_K1(K1v, CALL)                      \
  switch (K1v) {                                    \
    case 1: { constexpr int K1 = 1; CALL; } break;  \
    case 2: { constexpr int K1 = 2; CALL; } break;  \
    case 3: { constexpr int K1 = 3; CALL; } break;  \
    case 4: { constexpr int K1 = 4; CALL; } break;  \
    default: set_error("unsupported k"); return PINN_ERR_UNSUPPORTED; \
  }

static int prep(const Net& n, int64_t N, int64_t ws_bytes, void* ws, Layout* lo) {
  if (!make_layout(n, N, lo)) { set_error("too many layers for the generic engine"); return PINN_ERR_UNSUPPORTED; }
  if (!ws || ws_bytes < lo->total) {
    set_error("workspace too small: need %lld bytes, got %lld", (long long)lo->total, (long long)ws_bytes);
    return PINN_ERR_WORKSPACE;
  }
  return PINN_OK;
}

int generic_forward(const Net& n, const float* params, const float* X, int64_t N, float* Y, float* dY, void* ws,
                    int64_t ws_bytes, hipStream_t s) {
  Layout lo;
  int rc = prep(n, N, ws_bytes, ws, &lo);
  if (rc) return rc;
  const unsigned grid = (unsigned)lo.nblocks;
  DISPATCH_K1(n.K1, {
    rc = run_forward<K1>(n, params, X, N, (char*)ws, lo, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unseed<K1>, dim3(grid), dim3(TPB), 0, s, (const float*)((char*)ws + lo.act_off[n.L + 1]),
                       n.d_out, N, Y, dY);
  });
  return check_launch("generic unseed");
}

int generic_jet_backward(const Net& n, const float* params, const float* X, int64_t N, const float* gY,
                         const float* gdY, float* grad, void* ws, int64_t ws_bytes, hipStream_t s) {
  Layout lo;
  int rc = prep(n, N, ws_bytes, ws, &lo);
  if (rc) return rc;
  const unsigned grid = (unsigned)lo.nblocks;
  DISPATCH_K1(n.K1, {
    rc = run_forward<K1>(n, params, X, N, (char*)ws, lo, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_seed_adjoint<K1>, dim3(grid), dim3(TPB), 0, s, gY, gdY, n.d_out, N,
                       (float*)((char*)ws + lo.g0));
    rc = run_backward<K1>(n, params, N, (char*)ws, lo, grad, s);
  });
  return rc;
}

int generic_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N, void* ws,
                 int64_t ws_bytes, hipStream_t s) {
  Layout lo;
  int rc = prep(n, N, ws_bytes, ws, &lo);
  if (rc) return rc;
  DISPATCH_K1(n.K1, { rc = run_loss<K1>(n, rq, params, X, N, (char*)ws, lo, s); });
  return rc;
}

}  // namespace pinn
