// pinn_chain_w256.hip — bf16-mode chain kernels, padded hidden width 256 (see pinn_chain_launch.inc)
#define CHAIN_NTW 16
#include "pinn_chain_launch.inc"
