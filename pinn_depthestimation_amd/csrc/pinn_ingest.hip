// pinn_ingest.hip — the reference's collocation-point staging (train.py:246-277, operations.py:4-30) on the device:
// raw grids as scipy.io.loadmat yields them (float64, ny x nx row-major) -> the (N, d_in) fp32 matrix the hot
// path consumes.  Same arithmetic in the same order as the NumPy statements, in float64, cast to fp32 last
// (train.py:88 `.float()`), so the result equals the host path bit for bit:
//   operations.py:26-27   np.nanmin / np.nanmax of a variable                 -> pinn_nanminmax_f64
//   train.py:260          data[key][::interval_x, ::interval_y]               \
//   operations.py:4-8     2 * (data - min) / (max - min) - 1  (zeros if max == min)   |  pinn_stage_grid_columns
//   train.py:265-267      reshape / transpose / reshape(-1, 1): column-major flatten  |
//   train.py:276-277      drop rows with a NaN in any column, order kept       /
#include "common.h"

namespace pinn {
namespace {

constexpr int IB = 256;

__global__ void k_nanminmax_partial(const double* __restrict__ d, int64_t n, double* __restrict__ part) {
  __shared__ double smin[IB], smax[IB];
  double lo = INFINITY, hi = -INFINITY;     // np.nanmin of an all-NaN array is NaN (with a warning): handled in the final pass
  for (int64_t i = (int64_t)blockIdx.x * IB + threadIdx.x; i < n; i += (int64_t)gridDim.x * IB) {
    const double v = d[i];
    if (v == v) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = IB / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      smin[threadIdx.x] = smin[threadIdx.x + s] < smin[threadIdx.x] ? smin[threadIdx.x + s] : smin[threadIdx.x];
      smax[threadIdx.x] = smax[threadIdx.x + s] > smax[threadIdx.x] ? smax[threadIdx.x + s] : smax[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}
__global__ void k_nanminmax_final(const double* __restrict__ part, int nb, double* __restrict__ out2) {
  __shared__ double smin[IB], smax[IB];
  double lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += IB) {
    lo = part[2 * i] < lo ? part[2 * i] : lo;
    hi = part[2 * i + 1] > hi ? part[2 * i + 1] : hi;
  }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = IB / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      smin[threadIdx.x] = smin[threadIdx.x + s] < smin[threadIdx.x] ? smin[threadIdx.x + s] : smin[threadIdx.x];
      smax[threadIdx.x] = smax[threadIdx.x + s] > smax[threadIdx.x] ? smax[threadIdx.x + s] : smax[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool none = smin[0] > smax[0];                 // no finite-or-inf value at all: all NaN (or empty)
    out2[0] = none ? NAN : smin[0];
    out2[1] = none ? NAN : smax[0];
  }
}

struct Grids { const double* p[16]; };

// normalised value of variable c at flattened row r (column-major over the SUBSAMPLED grid: r = j * nys + i)
__device__ inline double staged(const Grids& g, const double* __restrict__ mm, int c, int64_t r, int64_t nys,
                                int64_t nx, int ix, int iy) {
#pragma clang fp contract(off)
  const int64_t j = r / nys, i = r % nys;
  const double v = g.p[c][(i * ix) * nx + j * iy];
  const double lo = mm[2 * c], hi = mm[2 * c + 1];
  if (hi == lo) return v == v ? 0.0 : 0.0;               // np.zeros_like(data): the NaN is gone too
  return 2 * (v - lo) / (hi - lo) - 1;
}

__global__ void k_stage_count(Grids g, int d_in, const double* __restrict__ mm, int64_t nys, int64_t nxs, int64_t nx,
                              int ix, int iy, int* __restrict__ block_cnt) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * IB + threadIdx.x;
  if (r < nys * nxs) {
    bool ok = true;
    for (int c = 0; c < d_in; ++c) { const double v = staged(g, mm, c, r, nys, nx, ix, iy); ok = ok && (v == v); }
    if (ok) atomicAdd(&cnt, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = cnt;
}
// exclusive scan of the per-block counts by ONE block (<= 2^31 rows / 256 = 8 M blocks; it is a staging step)
__global__ void k_stage_scan(int* __restrict__ block_cnt, int nb, int64_t* __restrict__ n_rows) {
  __shared__ int tmp[IB];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += IB) {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? block_cnt[i] : 0;
    tmp[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < IB; s <<= 1) {
      const int a = threadIdx.x >= s ? tmp[threadIdx.x - s] : 0;
      __syncthreads();
      tmp[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < nb) block_cnt[i] = carry + tmp[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += tmp[IB - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_rows = carry;
}
__global__ void k_stage_write(Grids g, int d_in, const double* __restrict__ mm, int64_t nys, int64_t nxs, int64_t nx,
                              int ix, int iy, const int* __restrict__ block_off, float* __restrict__ X) {
  __shared__ int flag[IB];
  const int64_t r = (int64_t)blockIdx.x * IB + threadIdx.x;
  double v[16];
  bool ok = r < nys * nxs;
  if (ok)
    for (int c = 0; c < d_in; ++c) { v[c] = staged(g, mm, c, r, nys, nx, ix, iy); ok = ok && (v[c] == v[c]); }
  flag[threadIdx.x] = ok ? 1 : 0;
  __syncthreads();
  int rank = 0;                                          // rows of this block in front of mine (order kept)
  for (int t = 0; t < (int)threadIdx.x; ++t) rank += flag[t];
  if (ok) {
    const int64_t row = (int64_t)block_off[blockIdx.x] + rank;
    for (int c = 0; c < d_in; ++c) X[row * d_in + c] = (float)v[c];
  }
}

}  // namespace
}  // namespace pinn

using namespace pinn;

extern "C" {

int32_t pinn_nanminmax_f64(const double* data, int64_t n, double* out2, void* ws, int64_t ws_bytes, void* stream) {
  if (!data || n < 0 || !out2) { set_error("NULL pointer argument"); return PINN_ERR_INVALID; }
  const int nb = (int)(n / IB + 1 < 1024 ? n / IB + 1 : 1024);
  if (!ws || ws_bytes < (int64_t)nb * 16) { set_error("workspace too small: need %d bytes", nb * 16); return PINN_ERR_WORKSPACE; }
  hipLaunchKernelGGL(k_nanminmax_partial, dim3(nb), dim3(IB), 0, (hipStream_t)stream, data, n, (double*)ws);
  hipLaunchKernelGGL(k_nanminmax_final, dim3(1), dim3(IB), 0, (hipStream_t)stream, (const double*)ws, nb, out2);
  return check_launch("nanminmax");
}

int64_t pinn_stage_workspace_bytes(int64_t ny, int64_t nx, int32_t ix, int32_t iy) {
  if (ny < 1 || nx < 1 || ix < 1 || iy < 1) return -1;
  const int64_t rows = ((ny + ix - 1) / ix) * ((nx + iy - 1) / iy);
  return ((rows + IB - 1) / IB + 1) * 4 + 256;
}

int32_t pinn_stage_grid_columns(const double* const* grids, int32_t d_in, int64_t ny, int64_t nx, int32_t ix, int32_t iy,
                                const double* minmax, float* X_out, int64_t* n_rows_out, void* ws, int64_t ws_bytes,
                                void* stream) {
  if (!grids || !minmax || !X_out || !n_rows_out) { set_error("NULL pointer argument"); return PINN_ERR_INVALID; }
  if (d_in < 1 || d_in > 16 || ny < 1 || nx < 1 || ix < 1 || iy < 1) { set_error("bad grid geometry"); return PINN_ERR_INVALID; }
  const int64_t nys = (ny + ix - 1) / ix, nxs = (nx + iy - 1) / iy, rows = nys * nxs;
  if (rows > ((int64_t)1 << 31) - 1) { set_error("more than 2^31 staged rows"); return PINN_ERR_UNSUPPORTED; }
  const int64_t need = pinn_stage_workspace_bytes(ny, nx, ix, iy);
  if (!ws || ws_bytes < need) { set_error("workspace too small: need %lld bytes", (long long)need); return PINN_ERR_WORKSPACE; }
  Grids g;
  for (int c = 0; c < 16; ++c) g.p[c] = c < d_in ? grids[c] : nullptr;
  for (int c = 0; c < d_in; ++c) if (!g.p[c]) { set_error("grid %d is NULL", c); return PINN_ERR_INVALID; }
  const int nb = (int)((rows + IB - 1) / IB);
  int* cnt = (int*)ws;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_stage_count, dim3(nb), dim3(IB), 0, s, g, d_in, minmax, nys, nxs, nx, ix, iy, cnt);
  hipLaunchKernelGGL(k_stage_scan, dim3(1), dim3(IB), 0, s, cnt, nb, n_rows_out);
  hipLaunchKernelGGL(k_stage_write, dim3(nb), dim3(IB), 0, s, g, d_in, minmax, nys, nxs, nx, ix, iy, (const int*)cnt, X_out);
  return check_launch("stage grid columns");
}

}  // extern "C"
