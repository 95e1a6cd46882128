// pinn_fused_batch_w32_k3.hip — batch kernel instances, padded hidden width 32, K1 = 3 (see pinn_fused_batch.inc)
#define BATCH_WP 32
#define BATCH_K1 3
#include "pinn_fused_batch.inc"
