// pinn_chain_w128.hip — bf16-mode chain kernels, padded hidden width 128 (see pinn_chain_launch.inc)
#define CHAIN_NTW 8
#include "pinn_chain_launch.inc"
