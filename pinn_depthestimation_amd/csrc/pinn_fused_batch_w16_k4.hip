// pinn_fused_batch_w16_k4.hip — batch kernel instances, padded hidden width 16, K1 = 4 (see pinn_fused_batch.inc)
#define BATCH_WP 16
#define BATCH_K1 4
#include "pinn_fused_batch.inc"
