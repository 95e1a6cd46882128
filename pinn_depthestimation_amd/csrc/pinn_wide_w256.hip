// pinn_wide_w256.hip — wide engine, padded width 256: fp32 kernels and the per-precision dispatchers
#define WIDE_NTW 16
#define WIDE_PART 0
#include "pinn_wide_launch.inc"
