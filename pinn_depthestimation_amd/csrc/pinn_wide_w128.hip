// pinn_wide_w128.hip — wide engine, padded width 128: fp32 kernels and the per-precision dispatchers
#define WIDE_NTW 8
#define WIDE_PART 0
#include "pinn_wide_launch.inc"
