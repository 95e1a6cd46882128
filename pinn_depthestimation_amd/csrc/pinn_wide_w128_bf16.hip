// pinn_wide_w128_bf16.hip — wide engine, padded width 128: the bf16-mode kernels (see pinn_wide_launch.inc)
#define WIDE_NTW 8
#define WIDE_PART 1
#include "pinn_wide_launch.inc"
