// common.h — shared host/device declarations for libpinn_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/pinn_hip.h"

namespace pinn {

void set_error(const char* fmt, ...);

// layer geometry helpers: layers = [d_in] + [width]*n_hidden + [d_out] (train.py:56)
constexpr int FUSED_KERNEL_AUTO = 0, FUSED_KERNEL_TILE = 1, FUSED_KERNEL_COOP = 2, FUSED_KERNEL_BATCH = 3;
struct Net {
  int d_in, d_out, L /*hidden layers*/, W, k, K1, act, prec;
  int fused_kernel;  // FUSED_KERNEL_*: which fused kernel desc.engine asked for (PINN_ENGINE_FUSED_TILE / _COOP / _BATCH)
  float drop_p;      // nn.Dropout rate in training mode (dnn.py:38), 0 = off
  uint32_t drop_seed, drop_thresh;   // keep unit iff dropout_bits(...) >= drop_thresh (= p * 2^32)
  int dir_col[PINN_MAX_DIRS];
  int n_lin;  // L + 1 linear layers
  __host__ __device__ int in_dim(int l) const { return l == 0 ? d_in : W; }
  __host__ __device__ int out_dim(int l) const { return l == L ? d_out : W; }
  __host__ __device__ int64_t w_off(int l) const {  // offset of W_l in the flat parameter vector (O(1): 100-layer nets)
    if (l == 0) return 0;
    const int64_t hid = (int64_t)d_in * W + W + (int64_t)((l <= L ? l : L) - 1) * ((int64_t)W * W + W);   // layer 0 + hidden blocks
    return l <= L ? hid : hid + (int64_t)W * d_out + d_out;                                        // l == L + 1: the total
  }
  __host__ __device__ int layer_of(int64_t i) const {   // which layer's block holds flat parameter index i
    const int64_t s0 = (int64_t)d_in * W + W, per = (int64_t)W * W + W;
    if (i < s0) return 0;
    const int64_t l = 1 + (i - s0) / per;
    return l < L ? (int)l : L;
  }
  __host__ __device__ int64_t b_off(int l) const { return w_off(l) + (int64_t)in_dim(l) * out_dim(l); }
  __host__ __device__ int64_t n_params() const { return w_off(L + 1); }
};

int make_net(const pinn_desc* d, Net* n);  // validates, returns PINN_OK or error

// Counter-based dropout mask (two rounds of the murmur3 finaliser over (seed, point) then (layer, feature)):
// 32 uniform bits per (seed, hidden layer, unit, point); the unit is kept iff bits >= p * 2^32.  Stateless, so
// the reverse sweep re-derives the forward's mask from the same seed (no mask storage, any launch geometry).
__host__ __device__ inline uint32_t dropout_fmix(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t dropout_bits(uint32_t seed, uint32_t layer, uint32_t feature, uint64_t point) {
  uint32_t x = dropout_fmix(seed ^ ((uint32_t)point * 0x9E3779B1u));
  x ^= (layer * 0x01000193u + feature) * 0x9E3779B1u + (uint32_t)(point >> 32);
  return dropout_fmix(x);
}
__host__ __device__ inline uint32_t dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}

// what a loss call asks the engines for
// torch.optim.Adam's update folded into the kernel that finishes a loss + gradient pass (pinn_loss_grad_adam_step)
struct AdamReq {
  float* params;        // updated in place (the pass itself read the PACKED copy of them in the workspace)
  float* m; float* v;
  float w1, b2, w2, eps, step_size, bc2_sqrt;   // as pinn_adam_step forms them
  bool packed_valid;    // the workspace's packed weights already equal params (left there by the previous call)
  int n_loss_rows; const float* loss_rows; float* losses;   // optional weighted loss values (see pinn_adam_state)
};

struct LossReq {
  int kind;  // 0 = residual, 1 = mse, 2 = residual + mse in one pass
  int64_t n_split;  // kind 2: < 0 both terms on every point; >= 0 residual on points [0, n_split), mse on [n_split, N)
  pinn_residual_spec spec;
  const float* scale;   // residual: device term_scale (may be null when !want_grad)
  float* sums;          // residual: device term_sums
  int n_terms;          // residual: number of terms
  // mse
  const float* T; int n_cols; int out_col[PINN_MAX_ROLES];
  const float* mse_scale;   // device col_scale
  float* mse_sums;          // device col_sums
  float* grad;          // device flat grad (+=) or null
  const AdamReq* adam;  // fused engine only: grad is OVERWRITTEN with this pass's gradient and the update applied
};

// generic engine (pinn_generic.hip)
int64_t generic_workspace_bytes(const Net& n, int64_t N);
int generic_forward(const Net& n, const float* params, const float* X, int64_t N, float* Y, float* dY,
                    void* ws, int64_t ws_bytes, hipStream_t s);
int generic_jet_backward(const Net& n, const float* params, const float* X, int64_t N, const float* gY,
                         const float* gdY, float* grad, void* ws, int64_t ws_bytes, hipStream_t s);
int generic_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N,
                 void* ws, int64_t ws_bytes, hipStream_t s);

// fused MFMA engine (pinn_fused.hip)
bool fused_supports(const Net& n, bool want_grad);
bool fused_supports_adam(const Net& n, const LossReq& rq, int64_t N);   // one-pass requests only
int64_t fused_workspace_bytes(const Net& n, int64_t N);
int fused_forward(const Net& n, const float* params, const float* X, int64_t N, float* Y, float* dY,
                  void* ws, int64_t ws_bytes, hipStream_t s);
int fused_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N,
               void* ws, int64_t ws_bytes, hipStream_t s);

// wide MFMA engine, 64 < W <= 256 (pinn_wide.hip)
bool wide_supports(const Net& n);
int64_t wide_workspace_bytes(const Net& n, int64_t N);
int wide_forward(const Net& n, const float* params, const float* X, int64_t N, float* Y, float* dY,
                 void* ws, int64_t ws_bytes, hipStream_t s);
int wide_loss(const Net& n, const LossReq& rq, const float* params, const float* X, int64_t N,
              void* ws, int64_t ws_bytes, hipStream_t s);

// compute units of the CURRENT device (cached per device ordinal: one process may drive several GPUs)
int device_cu_count();
// raise a kernel's dynamic-LDS limit above 64 KB, once per (kernel, device) instead of on every launch
int ensure_dynamic_lds(const void* kernel, size_t lds_bytes);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return PINN_ERR_LAUNCH;
  }
  return PINN_OK;
}

}  // namespace pinn
