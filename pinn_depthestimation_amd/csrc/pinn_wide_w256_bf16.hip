// pinn_wide_w256_bf16.hip — wide engine, padded width 256: the bf16-mode kernels (see pinn_wide_launch.inc)
#define WIDE_NTW 16
#define WIDE_PART 1
#include "pinn_wide_launch.inc"
