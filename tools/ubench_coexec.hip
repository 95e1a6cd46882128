// ubench_coexec.hip — does a gfx950 SIMD overlap one wave's VALU work with another wave's
// v_mfma_f32_16x16x4_f32 stream?  (design input for fused_pair_kernel.h)
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench_coexec.hip -o tools/ubench_coexec
//   run:   tools/ubench_coexec
// Each workgroup = WPS waves per SIMD (256*WPS threads), one workgroup per CU, ITERS iterations of
// [NM MFMAs on NACC accumulators] then [NV independent v_fma_f32], phases optionally separated by
// a scheduling barrier.  Prints cycles per iteration per SIMD (wall time * clock / iters).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NACC, bool SEP, bool TRANS>
__global__ __launch_bounds__(512) void ub(float* out, int iters, float x) {
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = x;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  if ((threadIdx.x >> 8) & 1) {   // the second wave of each SIMD starts half a period later
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i % 8] = fmaf(v[i % 8], b, a);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % NACC], 0, 0, 0);
    if (SEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (TRANS && (i % 8) == 0) v[i % 8] = __builtin_amdgcn_rcpf(v[i % 8]);
      else v[i % 8] = fmaf(v[i % 8], b, a);
    }
    if (SEP) __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave: every MFMA followed by K independent VALU ops (pinned with sched_group_barrier)
template <int NM, int K, int NACC>
__global__ __launch_bounds__(256) void ub_inter(float* out, int iters, float x) {
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = x;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % NACC], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[(i * K + k) % 8] = fmaf(v[(i * K + k) % 8], b, a);
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// two waves per SIMD with fixed roles: waves 0-3 only MFMA, waves 4-7 only VALU; each role's own cycle count
template <int NM, int NV>
__global__ __launch_bounds__(512) void ub_roles(float* out, long long* cyc, int iters, float x, int mode) {
  f4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = x;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  const bool second = (threadIdx.x >> 8) & 1;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (!second) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i % 4], 0, 0, 0);
      }
  } else {
    if (mode & 4) __builtin_amdgcn_s_setprio(3);   // let the vector wave win issue arbitration
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i % 8] = fmaf(v[i % 8], b, a);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

typedef short bf16x4 __attribute__((ext_vector_type(4)));
// same role split with bf16 MFMAs (a separate datapath from the fp32 vector ALUs)
template <int NM, int NV>
__global__ __launch_bounds__(512) void ub_roles_bf16(float* out, long long* cyc, int iters, float x, int mode) {
  f4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = x;
  bf16x4 av = {(short)threadIdx.x, 1, 2, 3}, bv = {3, 2, 1, (short)threadIdx.x};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  const bool second = (threadIdx.x >> 8) & 1;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (!second) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i % 4] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc[i % 4], 0, 0, 0);
      }
  } else {
    if (mode & 4) __builtin_amdgcn_s_setprio(3);   // let the vector wave win issue arbitration
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i % 8] = fmaf(v[i % 8], b, a);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NM, int NV>
static void run_roles_bf16(float* out, long long* cyc, int mode) {
  const int iters = 2000;
  hipLaunchKernelGGL((ub_roles_bf16<NM, NV>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0001f, mode);
  (void)hipDeviceSynchronize();
  long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("bf16 roles mode=%d  MFMA(16x16x16 bf16) wave: %8.1f ticks per %d MFMAs;  VALU wave: %8.1f ticks per %d v_fma\n",
         mode, (double)h[0] / iters, NM, (double)h[4] / iters, NV);
}

template <int NM, int K, int NACC>
static void run_inter(float* out, double ghz) {
  const int iters = 2000, grid = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((ub_inter<NM, K, NACC>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((ub_inter<NM, K, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("one wave, each MFMA followed by %d v_fma (pinned)   %8.1f cycles per MFMA+VALU group (MFMA alone ~34)\n", K,
         ms * 1e-3 * ghz * 1e9 / iters / NM);
}

template <int NM, int NV>
static void run_roles(float* out, long long* cyc, int mode) {
  const int iters = 2000;
  hipLaunchKernelGGL((ub_roles<NM, NV>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0001f, mode);
  (void)hipDeviceSynchronize();
  long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // s_memtime ticks (constant-rate counter): compare modes, not absolute cycles
  printf("fp32 roles mode=%d (1: MFMA waves run, 2: VALU waves run)  MFMA(16x16x4 f32) wave: %8.1f ticks per %d MFMAs;  VALU wave: %8.1f ticks per %d v_fma\n",
         mode, (double)h[0] / iters, NM, (double)h[4] / iters, NV);
}

template <int NM, int NV, int NACC, bool SEP, bool TRANS>
static void run(const char* name, int wps, float* out, double ghz) {
  const int iters = 2000, grid = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((ub<NM, NV, NACC, SEP, TRANS>), dim3(grid), dim3(256 * wps), 0, 0, out, 10, 1.0001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((ub<NM, NV, NACC, SEP, TRANS>), dim3(grid), dim3(256 * wps), 0, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * ghz * 1e9 / iters;
  printf("%-44s wps=%d  NM=%3d NV=%3d NACC=%d  %8.1f cycles/iter/SIMD   (MFMA-only ideal %5d, VALU-only ideal %5d per wave)\n",
         name, wps, NM, NV, NACC, cyc, NM * 32, NV * 4);
}

// ---- round 2: ONE wave, K hand-pinned fillers behind every MFMA, for the MFMA forms whose gap is longer than the
// fp32 16x16x4's (guide rows 'vector-instruction ISSUE cost' / 'single-issue instructions HIDDEN').
//   KIND 0: v_mfma_f32_16x16x4_f32   (32-cycle issue, the form k_fused uses)
//   KIND 1: v_mfma_f32_32x32x2_f32   (64-cycle issue, same flop rate)
//   KIND 2: v_mfma_f32_16x16x32_bf16 (16-cycle issue, the form the chain kernels use)
//   KIND 3: v_mfma_f32_32x32x16_bf16 (32-cycle issue)
// TRANS: every filler is a v_exp_f32 (8-cycle issue) instead of a v_fma_f32 (4).
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
template <int KIND> struct Acc { typedef f4 type; };
template <> struct Acc<1> { typedef f16v type; };
template <> struct Acc<3> { typedef f16v type; };
template <int KIND>
__device__ __forceinline__ typename Acc<KIND>::type mf(float a, float b, bf8v av, bf8v bv, typename Acc<KIND>::type c) {
  if constexpr (KIND == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  else if constexpr (KIND == 1) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  else if constexpr (KIND == 2) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
}
template <int KIND, int NM, int K, bool TRANS>
__global__ __launch_bounds__(256) void ub_inter2(float* out, int iters, float x) {
  constexpr int NACC = 4;
  typename Acc<KIND>::type acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < (int)(sizeof(acc[0]) / 4); ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = x;
  bf8v av, bv;
#pragma unroll
  for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j); bv[j] = (__bf16)(b - j); }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      acc[i % NACC] = mf<KIND>(a, b, av, bv, acc[i % NACC]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float& t = v[(i * K + k) % 8];
        t = TRANS ? __builtin_amdgcn_exp2f(t) : fmaf(t, b, a);
      }
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (K > 0) __builtin_amdgcn_sched_group_barrier(TRANS ? 0x400 : 0x002, K, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < (int)(sizeof(acc[0]) / 4); ++j) s += acc[i][j];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int K, bool TRANS>
static double time_inter2(float* out, double ghz) {
  constexpr int NM = 64;
  const int iters = 2000, grid = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((ub_inter2<KIND, NM, K, TRANS>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((ub_inter2<KIND, NM, K, TRANS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * ghz * 1e9 / iters / NM;
}
template <int KIND, bool TRANS>
static void table_inter2(const char* name, int gap, float* out, double ghz) {
  printf("%-28s gap %2d | %s fillers per MFMA: K=0 %6.1f  K=1 %6.1f  K=2 %6.1f  K=3 %6.1f  K=4 %6.1f  K=5 %6.1f  K=6 %6.1f  K=8 %6.1f  cycles per MFMA+fillers\n",
         name, gap, TRANS ? "v_exp_f32" : "v_fma_f32", time_inter2<KIND, 0, TRANS>(out, ghz), time_inter2<KIND, 1, TRANS>(out, ghz),
         time_inter2<KIND, 2, TRANS>(out, ghz), time_inter2<KIND, 3, TRANS>(out, ghz), time_inter2<KIND, 4, TRANS>(out, ghz),
         time_inter2<KIND, 5, TRANS>(out, ghz), time_inter2<KIND, 6, TRANS>(out, ghz), time_inter2<KIND, 8, TRANS>(out, ghz));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  int khz = 0;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double ghz = khz * 1e-6;
  printf("clock %.3f GHz\n", ghz);
  run<128, 0, 4, true, false>("mfma only, 4 acc", 1, out, ghz);
  run<128, 0, 2, true, false>("mfma only, 2 acc", 1, out, ghz);
  run<128, 0, 1, true, false>("mfma only, 1 acc", 1, out, ghz);
  run<128, 0, 2, true, false>("mfma only, 2 acc", 2, out, ghz);
  run<0, 512, 1, true, false>("valu only", 1, out, ghz);
  run<0, 512, 1, true, false>("valu only", 2, out, ghz);
  run<0, 512, 1, true, true>("valu only (1/8 rcp)", 1, out, ghz);
  run<128, 512, 4, true, false>("mfma then valu, separated", 1, out, ghz);
  run<128, 512, 4, false, false>("mfma + valu, compiler-interleaved", 1, out, ghz);
  run<128, 512, 4, true, false>("mfma then valu, separated", 2, out, ghz);
  run<128, 512, 2, true, false>("mfma then valu, separated, 2 acc", 2, out, ghz);
  run<128, 1024, 4, true, false>("mfma then 2x valu, separated", 2, out, ghz);
  run<128, 512, 4, true, true>("mfma then valu(1/8 rcp), separated", 2, out, ghz);
  run<128, 512, 4, false, false>("mfma + valu, compiler-interleaved", 2, out, ghz);
  run_inter<128, 0, 4>(out, ghz);
  run_inter<128, 2, 4>(out, ghz);
  run_inter<128, 4, 4>(out, ghz);
  run_inter<128, 6, 4>(out, ghz);
  run_inter<128, 8, 4>(out, ghz);
  printf("---- one wave per SIMD, K fillers pinned behind every MFMA (host-timed at the nominal clock; compare rows, the chip clocks down under load) ----\n");
  table_inter2<0, false>("v_mfma_f32_16x16x4_f32", 32, out, ghz);
  table_inter2<1, false>("v_mfma_f32_32x32x2_f32", 64, out, ghz);
  table_inter2<2, false>("v_mfma_f32_16x16x32_bf16", 16, out, ghz);
  table_inter2<3, false>("v_mfma_f32_32x32x16_bf16", 32, out, ghz);
  table_inter2<0, true>("v_mfma_f32_16x16x4_f32", 32, out, ghz);
  table_inter2<1, true>("v_mfma_f32_32x32x2_f32", 64, out, ghz);
  table_inter2<3, true>("v_mfma_f32_32x32x16_bf16", 32, out, ghz);
  long long* cyc;
  (void)hipMalloc(&cyc, 8 * sizeof(long long));
  run_roles<128, 512>(out, cyc, 1);
  run_roles<128, 512>(out, cyc, 2);
  run_roles<128, 512>(out, cyc, 3);
  run_roles<128, 512>(out, cyc, 7);
  run_roles_bf16<128, 512>(out, cyc, 1);
  run_roles_bf16<128, 512>(out, cyc, 2);
  run_roles_bf16<128, 512>(out, cyc, 3);
  run_roles_bf16<128, 512>(out, cyc, 7);
  (void)hipFree(cyc);
  (void)hipFree(out);
  return 0;
}
