// ubench_ldsdma.hip — how fast can one workgroup (4 waves, one per SIMD) pull an L2-resident table into LDS with
// global_load_lds_dwordx4 (1 KB per wave-instruction), the way the chain kernels stream their weight slabs?
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench_ldsdma.hip -o tools/ubench_ldsdma
// Every workgroup sweeps the SAME `table_kb` KB table `sweeps` times in 16 KB slabs (4 x 1 KB per wave), keeping
// `inflight` slabs outstanding per wave (counted vmcnt) — optionally with a workgroup barrier per slab.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int INFLIGHT, bool BARRIER, int AUX>
__global__ __launch_bounds__(256, 1) void k(const char* table, int table_kb, int sweeps, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nslab = table_kb / 16;
  const long long t0 = __builtin_readcyclecounter();
  long long issued = 0, total = (long long)nslab * sweeps;
  auto issue = [&]() {
    if (issued >= total) return;
    const char* src = table + (issued % nslab) * 16384 + wave * 4096 + lane * 16;
    char* dst = smem + (issued % 8) * 16384 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 1024),
                                       (void __attribute__((address_space(3)))*)(dst + i * 1024), 16, 0, AUX);
    ++issued;
  };
  for (int i = 0; i < INFLIGHT; ++i) issue();
  for (long long g = 0; g < total; ++g) {
    if (g + INFLIGHT >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFLIGHT - 1) * 4) : "memory");
    if (BARRIER) __builtin_amdgcn_s_barrier();
    issue();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the chain kernels' GEMM step in miniature: wait for slab g, barrier, issue slab g + INFLIGHT, read the slab's 16
// fragments from LDS (ds_read_b128) and run NMFMA v_mfma_f32_16x16x32_bf16 on them
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int INFLIGHT, int NMFMA, int MODE>
__global__ __launch_bounds__(256, 1) void kstep2(const char* table, int table_kb, int sweeps, long long* cyc, float* out) {
  // MODE 0: all 16 fragment reads of a step issued up front (64 VGPRs), MFMAs behind them with counted lgkmcnt waits;
  // MODE 1: the same plus sched_group_barrier pinning (16 DS reads, then per fragment 4 MFMAs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nslab = table_kb / 16;
  long long issued = 0, total = (long long)nslab * sweeps;
  auto issue = [&]() {
    if (issued >= total) return;
    const char* src = table + (issued % nslab) * 16384 + wave * 4096 + lane * 16;
    char* dst = smem + (issued % 8) * 16384 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 1024),
                                       (void __attribute__((address_space(3)))*)(dst + i * 1024), 16, 0, 0);
    ++issued;
  };
  f4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
  bf8 b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)(lane + i + j);
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < INFLIGHT; ++i) issue();
  for (long long g = 0; g < total; ++g) {
    if (g + INFLIGHT >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFLIGHT - 1) * 4) : "memory");
    if (MODE != 3) __builtin_amdgcn_s_barrier();
    if (MODE < 2) issue();                             // MODE 2: no copies after the prologue (reads see stale LDS: timing only); MODE 3: no barrier either
    if (MODE == 4 && issued < total) {                 // MODE 4: the copies as buffer_load ... lds (scalar resource + 32-bit lane offset)
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(table + (issued % nslab) * 16384 + wave * 4096), 0, 4096, 0x00020000);
      char* dst = smem + (issued % 8) * 16384 + wave * 4096;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(dst), 16, lane * 16, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(dst + 1024), 16, lane * 16, 0, 1024, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(dst + 2048), 16, lane * 16, 0, 2048, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(dst + 3072), 16, lane * 16, 0, 3072, 0);
      ++issued;
    }
    const char* sl = smem + (g % 8) * 16384 + lane * 16;
    bf8 fr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) fr[s] = *reinterpret_cast<const bf8*>(sl + s * 1024);
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int c = 0; c < NMFMA / 16; ++c) acc[(s * 4 + c) % 16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[s], b[c % 4], acc[(s * 4 + c) % 16], 0, 0, 0);
    if (MODE >= 1) {
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / 16, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sacc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the step with REGISTER staging instead of LDS-DMA: each wave loads its quarter of a slab (4 x 1 KB) into 16 VGPRs with
// global_load_dwordx4 two steps ahead and ds_write_b128's it into the ring one step ahead of its use
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void kstep_reg(const char* table, int table_kb, int sweeps, long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nslab = table_kb / 16;
  const long long total = (long long)nslab * sweeps;
  f4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
  bf8 b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)(lane + i + j);
  u4v st[4];
  auto gload = [&](long long g2) {
    const char* src = table + ((g2 < total ? g2 : total - 1) % nslab) * 16384 + wave * 4096 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = *reinterpret_cast<const u4v*>(src + i * 1024);
  };
  auto lwrite = [&](long long g2) {
    char* dst = smem + (g2 % 4) * 16384 + wave * 4096 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u4v*>(dst + i * 1024) = st[i];
  };
  const long long t0 = __builtin_readcyclecounter();
  gload(0); lwrite(0); gload(1);
  __syncthreads();
  for (long long g = 0; g < total; ++g) {
    // registers hold slab g + 1 (requested one step ago): into the ring, then request slab g + 2
    lwrite(g + 1);
    gload(g + 2);
    const char* sl = smem + (g % 4) * 16384 + lane * 16;
    bf8 fr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) fr[s] = *reinterpret_cast<const bf8*>(sl + s * 1024);
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[(s * 4 + c) % 16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[s], b[c % 4], acc[(s * 4 + c) % 16], 0, 0, 0);
    if (MODE == 1) {
      __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);   // DS writes
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // VMEM reads
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __syncthreads();      // slab g + 1 is in LDS for everyone; everyone is done reading slab g
  }
  const long long t1 = __builtin_readcyclecounter();
  float sacc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc + (float)st[0][0];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run_step_reg(const char* name, const char* table, long long* cyc, float* out, int grid) {
  const int sweeps = 40, table_kb = 2816;
  hipFuncSetAttribute((const void*)kstep_reg<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
  hipLaunchKernelGGL((kstep_reg<MODE>), dim3(grid), dim3(256), 4 * 16384, 0, table, table_kb, 2, cyc, out);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((kstep_reg<MODE>), dim3(grid), dim3(256), 4 * 16384, 0, table, table_kb, sweeps, cyc, out);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
  printf("%-44s grid %3d            : %7.1f cycles per 16 KB step (64 MFMA = 1024 cycles of matrix pipe)\n", name, grid,
         avg / ((double)(table_kb / 16) * sweeps));
}

// the same step on v_mfma_f32_32x32x16_bf16 (32-cycle issue, ~3 single-issue instructions hide per MFMA — ubench_coexec):
// 16 fragments x 2 column groups = 32 MFMAs = the same 1024 matrix-pipe cycles; copies and LDS reads pinned between them
typedef float f16v __attribute__((ext_vector_type(16)));
template <int INFLIGHT, int MODE>
__global__ __launch_bounds__(256, 1) void kstep32(const char* table, int table_kb, int sweeps, long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nslab = table_kb / 16;
  long long issued = 0, total = (long long)nslab * sweeps;
  f16v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf8 b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)(lane + i + j);
  auto issue1 = [&](long long g2, int i) {
    if (g2 >= total) return;
    const char* src = table + (g2 % nslab) * 16384 + wave * 4096 + lane * 16;
    char* dst = smem + (g2 % 8) * 16384 + wave * 4096;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 1024),
                                     (void __attribute__((address_space(3)))*)(dst + i * 1024), 16, 0, 0);
  };
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < INFLIGHT; ++i) { for (int j = 0; j < 4; ++j) issue1(issued, j); ++issued; }
  for (long long g = 0; g < total; ++g) {
    if (g + INFLIGHT >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFLIGHT - 1) * 4) : "memory");
    __builtin_amdgcn_s_barrier();
    const char* sl = smem + (g % 8) * 16384 + lane * 16;
    bf8 fr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) fr[s] = *reinterpret_cast<const bf8*>(sl + s * 1024);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[(2 * s) % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], b[0], acc[(2 * s) % 8], 0, 0, 0);
      acc[(2 * s + 1) % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], b[1], acc[(2 * s + 1) % 8], 0, 0, 0);
      if (MODE == 1 && (s % 4) == 1) issue1(issued, s / 4);     // one copy behind every 8th MFMA
    }
    if (MODE == 0) { for (int j = 0; j < 4; ++j) issue1(issued, j); }
    ++issued;
    // schedule: 4 reads first, then per fragment 2 MFMAs + the next read; copies where the source put them
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (s < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (MODE == 1 && (s % 4) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sacc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) sacc += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int INFLIGHT, int MODE>
static void run_step32(const char* name, const char* table, long long* cyc, float* out, int grid) {
  const int sweeps = 40, table_kb = 2816;
  hipFuncSetAttribute((const void*)kstep32<INFLIGHT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
  hipLaunchKernelGGL((kstep32<INFLIGHT, MODE>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, 2, cyc, out);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((kstep32<INFLIGHT, MODE>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, sweeps, cyc, out);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
  printf("%-44s grid %3d inflight %d: %7.1f cycles per 16 KB step (32 MFMA 32x32x16 = 1024 cycles of matrix pipe)\n", name, grid, INFLIGHT,
         avg / ((double)(table_kb / 16) * sweeps));
}

template <int INFLIGHT, int NMFMA, int MODE>
static void run_step2(const char* name, const char* table, long long* cyc, float* out, int grid) {
  const int sweeps = 40, table_kb = 2816;
  hipFuncSetAttribute((const void*)kstep2<INFLIGHT, NMFMA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
  hipLaunchKernelGGL((kstep2<INFLIGHT, NMFMA, MODE>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, 2, cyc, out);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((kstep2<INFLIGHT, NMFMA, MODE>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, sweeps, cyc, out);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
  printf("%-44s grid %3d inflight %d: %7.1f cycles per 16 KB step (%d MFMA = %d cycles of matrix pipe)\n", name, grid, INFLIGHT,
         avg / ((double)(table_kb / 16) * sweeps), NMFMA, NMFMA * 16);
}

template <int INFLIGHT, int NMFMA, bool READS>
__global__ __launch_bounds__(256, 1) void kstep(const char* table, int table_kb, int sweeps, long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nslab = table_kb / 16;
  long long issued = 0, total = (long long)nslab * sweeps;
  auto issue = [&]() {
    if (issued >= total) return;
    const char* src = table + (issued % nslab) * 16384 + wave * 4096 + lane * 16;
    char* dst = smem + (issued % 8) * 16384 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 1024),
                                       (void __attribute__((address_space(3)))*)(dst + i * 1024), 16, 0, 0);
    ++issued;
  };
  f4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
  bf8 b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)(lane + i + j);
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < INFLIGHT; ++i) issue();
  for (long long g = 0; g < total; ++g) {
    if (g + INFLIGHT >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((INFLIGHT - 1) * 4) : "memory");
    __builtin_amdgcn_s_barrier();
    issue();
    const char* sl = smem + (g % 8) * 16384 + lane * 16;
    if (READS) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const bf8 a = *reinterpret_cast<const bf8*>(sl + s * 1024);
        if (NMFMA > 0) {
#pragma unroll
          for (int c = 0; c < NMFMA / 16; ++c) acc[(s * 4 + c) % 16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[c % 4], acc[(s * 4 + c) % 16], 0, 0, 0);
        } else asm volatile("" ::"v"(a));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sacc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int INFLIGHT, int NMFMA, bool READS>
static void run_step(const char* name, const char* table, long long* cyc, float* out, int grid) {
  const int sweeps = 40, table_kb = 2816;
  hipFuncSetAttribute((const void*)kstep<INFLIGHT, NMFMA, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
  hipLaunchKernelGGL((kstep<INFLIGHT, NMFMA, READS>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, 2, cyc, out);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((kstep<INFLIGHT, NMFMA, READS>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, sweeps, cyc, out);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
  printf("%-44s grid %3d inflight %d: %7.1f cycles per 16 KB step (%d MFMA = %d cycles of matrix pipe)\n", name, grid, INFLIGHT,
         avg / ((double)(table_kb / 16) * sweeps), NMFMA, NMFMA * 16);
}

template <int INFLIGHT, bool BARRIER, int AUX>
static void run(const char* name, const char* table, int table_kb, long long* cyc, int grid) {
  const int sweeps = 40;
  hipFuncSetAttribute((const void*)k<INFLIGHT, BARRIER, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
  hipLaunchKernelGGL((k<INFLIGHT, BARRIER, AUX>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, 2, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<INFLIGHT, BARRIER, AUX>), dim3(grid), dim3(256), 8 * 16384, 0, table, table_kb, sweeps, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[1024]; hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
  const double bytes = (double)table_kb * 1024 * sweeps;
  printf("%-34s table %5d KB grid %3d inflight %d barrier %d aux %d: %6.1f B/clk/CU (s_memtime)  %6.1f GB/s/CU  chip %5.2f TB/s (wall)\n",
         name, table_kb, grid, INFLIGHT, (int)BARRIER, AUX, bytes / avg, bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12);
}

int main() {
  char* table; hipMalloc(&table, 64 << 20); hipMemset(table, 1, 64 << 20);
  long long* cyc; hipMalloc(&cyc, 1024 * sizeof(long long));
  for (int grid : {1, 32, 256}) {
    run<2, false, 0>("L2-resident, 2 slabs ahead", table, 2816, cyc, grid);
    run<5, false, 0>("L2-resident, 5 slabs ahead", table, 2816, cyc, grid);
    run<7, false, 0>("L2-resident, 7 slabs ahead", table, 2816, cyc, grid);
    run<5, true, 0>("L2-resident, 5 ahead + barrier", table, 2816, cyc, grid);
    run<5, false, 0>("small table (256 KB)", table, 256, cyc, grid);
    run<5, false, 0>("beyond L2 (32 MB: MALL)", table, 32768, cyc, grid);
  }
  float* out; hipMalloc(&out, 256 * 256 * 4);
  for (int grid : {1, 256}) {
    run_step<5, 0, false>("step: wait + barrier + 4 copies", table, cyc, out, grid);
    run_step<5, 0, true>("step: + 16 ds_read_b128", table, cyc, out, grid);
    run_step<5, 64, true>("step: + 16 ds_read_b128 + 64 MFMA", table, cyc, out, grid);
    run_step<7, 64, true>("step: + 16 ds_read_b128 + 64 MFMA", table, cyc, out, grid);
    run_step<2, 64, true>("step: + 16 ds_read_b128 + 64 MFMA", table, cyc, out, grid);
    run_step2<5, 64, 0>("step2: 16 reads up front, then 64 MFMA", table, cyc, out, grid);
    run_step2<5, 64, 1>("step2: same, sched_group_barrier pinned", table, cyc, out, grid);
    run_step2<5, 64, 2>("step2: pinned, NO copies (barrier kept)", table, cyc, out, grid);
    run_step2<5, 64, 3>("step2: pinned, no copies, no barrier", table, cyc, out, grid);
    run_step2<5, 64, 4>("step2: pinned, copies as buffer_load..lds", table, cyc, out, grid);
    run_step_reg<0>("step_reg: global_load + ds_write staging", table, cyc, out, grid);
    run_step_reg<1>("step_reg: same, pinned schedule", table, cyc, out, grid);
    run_step32<5, 0>("step32: 32x32x16, copies at the end", table, cyc, out, grid);
    run_step32<5, 1>("step32: 32x32x16, copies between MFMAs", table, cyc, out, grid);
  }
  return 0;
}
