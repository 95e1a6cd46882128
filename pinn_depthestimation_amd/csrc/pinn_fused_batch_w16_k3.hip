// pinn_fused_batch_w16_k3.hip — batch kernel instances, padded hidden width 16, K1 = 3 (see pinn_fused_batch.inc)
#define BATCH_WP 16
#define BATCH_K1 3
#include "pinn_fused_batch.inc"
