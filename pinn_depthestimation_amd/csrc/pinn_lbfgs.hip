// pinn_lbfgs.hip — the L-BFGS two-loop recursion (torch.optim.LBFGS, the optimiser of train.py:116-125,200)
// as six launches.  Python side and derivation: pinn_depthestimation_amd/lbfgs.py.
//
// History: S, Y are (m x P) row-major RINGS — logical pair i (0 = oldest) lives in physical row
// (head + i) % m — and M[pi][pj] = s_pi . y_pj (fp64, physical indices).  With q0 = -g:
//     triu(M) al = S q0 ;  q = q0 - Y^T al ;  tril(M^T) w = diag(M) al - H (Y q) ;  d = H q + S^T w
// (triangles taken in LOGICAL order).  Every reference config runs 50 000 L-BFGS iterations with
// history 100; torch's Python loops cost ~400 launches per iteration.
#include "common.h"

namespace pinn {
namespace {

constexpr int LB_T = 256;

// out[row] = sign * sum_e A[row][e] * x[e]   (one workgroup per row; fp64 combine)
__global__ void k_lb_rowdots(const float* __restrict__ A, const float* __restrict__ x, double sign, int64_t P,
                             double* __restrict__ out) {
  const float* a = A + (int64_t)blockIdx.x * P;
  float acc = 0.f;
  for (int64_t e = threadIdx.x; e < P; e += LB_T) acc = fmaf(a[e], x[e], acc);
  __shared__ double red[LB_T];
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int s = LB_T / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sign * red[0];
}

// M[slot][j] = s . Y_j ;  M[j][slot] = S_j . y   for every physical row j (rows `slot` already hold s, y)
__global__ void k_lb_push_dots(const float* __restrict__ S, const float* __restrict__ Y, const float* __restrict__ s,
                               const float* __restrict__ y, int slot, int m, int64_t P, double* __restrict__ M) {
  const int j = blockIdx.x;
  const float* Sj = S + (int64_t)j * P;
  const float* Yj = Y + (int64_t)j * P;
  float a0 = 0.f, a1 = 0.f;
  for (int64_t e = threadIdx.x; e < P; e += LB_T) {
    a0 = fmaf(s[e], Yj[e], a0);
    a1 = fmaf(Sj[e], y[e], a1);
  }
  __shared__ double r0[LB_T], r1[LB_T];
  r0[threadIdx.x] = (double)a0; r1[threadIdx.x] = (double)a1;
  __syncthreads();
  for (int t = LB_T / 2; t > 0; t >>= 1) {
    if (threadIdx.x < t) { r0[threadIdx.x] += r0[threadIdx.x + t]; r1[threadIdx.x] += r1[threadIdx.x + t]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    M[(int64_t)slot * m + j] = r0[0];
    M[(int64_t)j * m + slot] = r1[0];
  }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}

// one wave: al (logical i = k-1 .. 0):  al_i = (b_i - sum_{j>i} al_j M[pi][pj]) / M[pi][pi]
__global__ void k_lb_solve_upper(const double* __restrict__ M, const double* __restrict__ b, int head, int k, int m,
                                 double* __restrict__ al, float* __restrict__ alf) {
  const int lane = threadIdx.x;
  for (int j = lane; j < m; j += 64) { al[j] = 0.0; alf[j] = 0.f; }
  __syncthreads();
  for (int i = k - 1; i >= 0; --i) {
    const int pi = (head + i) % m;
    double part = 0.0;
    for (int j = i + 1 + lane; j < k; j += 64) {
      const int pj = (head + j) % m;
      part += al[pj] * M[(int64_t)pi * m + pj];
    }
    const double s = wave_sum(part);
    if (lane == 0) {
      const double v = (b[pi] - s) / M[(int64_t)pi * m + pi];
      al[pi] = v; alf[pi] = (float)v;
    }
    __syncthreads();
  }
}

// one wave: w (logical i = 0 .. k-1):  w_i = (M_ii al_i - H c_i - sum_{j<i} w_j M[pj][pi]) / M_ii
__global__ void k_lb_solve_lower(const double* __restrict__ M, const double* __restrict__ al, const double* __restrict__ c,
                                 double H, int head, int k, int m, double* __restrict__ w, float* __restrict__ wf) {
  const int lane = threadIdx.x;
  for (int j = lane; j < m; j += 64) { w[j] = 0.0; wf[j] = 0.f; }
  __syncthreads();
  for (int i = 0; i < k; ++i) {
    const int pi = (head + i) % m;
    double part = 0.0;
    for (int j = lane; j < i; j += 64) {
      const int pj = (head + j) % m;
      part += w[pj] * M[(int64_t)pj * m + pi];
    }
    const double s = wave_sum(part);
    if (lane == 0) {
      const double mii = M[(int64_t)pi * m + pi];
      const double v = (mii * al[pi] - H * c[pi] - s) / mii;
      w[pi] = v; wf[pi] = (float)v;
    }
    __syncthreads();
  }
}

// out[e] = alpha * base[e] + sign * sum_row coef[row] * A[row][e]    (rows with coef == 0 are unused slots)
__global__ void k_lb_combine(const float* __restrict__ base, float alpha, const float* __restrict__ A,
                             const float* __restrict__ coef, float sign, int m, int64_t P, float* __restrict__ out) {
  __shared__ float cf[256];
  for (int j = threadIdx.x; j < m; j += LB_T) cf[j] = coef[j];
  __syncthreads();
  const int64_t e = (int64_t)blockIdx.x * LB_T + threadIdx.x;
  if (e >= P) return;
  float acc = 0.f;
  for (int r = 0; r < m; ++r) acc = fmaf(cf[r], A[(int64_t)r * P + e], acc);
  out[e] = fmaf(alpha, base[e], sign * acc);
}

}  // namespace
}  // namespace pinn

using namespace pinn;

extern "C" {

int32_t pinn_lbfgs_push(float* S, float* Y, double* M, int32_t m, int64_t P, int32_t slot, const float* s,
                        const float* y, void* stream) {
  if (!S || !Y || !M || !s || !y || m < 1 || m > 256 || P < 1 || slot < 0 || slot >= m) {
    set_error("pinn_lbfgs_push: bad arguments"); return PINN_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(S + (int64_t)slot * P, s, (size_t)P * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemcpyAsync(Y + (int64_t)slot * P, y, (size_t)P * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("pinn_lbfgs_push: copy failed"); return PINN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_lb_push_dots, dim3(m), dim3(LB_T), 0, st, (const float*)S, (const float*)Y, s, y, slot, m, P, M);
  return check_launch("lbfgs push");
}

int32_t pinn_lbfgs_direction(const float* S, const float* Y, const double* M, int32_t m, int64_t P, int32_t head,
                             int32_t k, const float* g, double H, float* d, double* tmp /*4m*/, float* coef /*2m*/,
                             float* q /*P*/, void* stream) {
  if (!S || !Y || !M || !g || !d || !tmp || !coef || !q || m < 1 || m > 256 || P < 1 || k < 1 || k > m || head < 0 ||
      head >= m) {
    set_error("pinn_lbfgs_direction: bad arguments"); return PINN_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  double* b = tmp; double* al = tmp + m; double* c = tmp + 2 * m; double* w = tmp + 3 * m;
  float* alf = coef; float* wf = coef + m;
  const unsigned gp = (unsigned)((P + LB_T - 1) / LB_T);
  hipLaunchKernelGGL(k_lb_rowdots, dim3(m), dim3(LB_T), 0, st, S, g, -1.0, P, b);                 // b = S q0
  hipLaunchKernelGGL(k_lb_solve_upper, dim3(1), dim3(64), 0, st, M, (const double*)b, head, k, m, al, alf);
  hipLaunchKernelGGL(k_lb_combine, dim3(gp), dim3(LB_T), 0, st, g, -1.f, Y, (const float*)alf, -1.f, m, P, q);   // q = -g - Y^T al
  hipLaunchKernelGGL(k_lb_rowdots, dim3(m), dim3(LB_T), 0, st, Y, (const float*)q, 1.0, P, c);   // c = Y q
  hipLaunchKernelGGL(k_lb_solve_lower, dim3(1), dim3(64), 0, st, M, (const double*)al, (const double*)c, H, head, k, m, w, wf);
  hipLaunchKernelGGL(k_lb_combine, dim3(gp), dim3(LB_T), 0, st, (const float*)q, (float)H, S, (const float*)wf, 1.f, m, P, d);  // d = H q + S^T w
  return check_launch("lbfgs direction");
}

}  // extern "C"
