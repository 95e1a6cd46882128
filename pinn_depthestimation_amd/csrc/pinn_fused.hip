// pinn_fused.hip — placeholder until the MFMA chain engine lands
#include "common.h"
namespace pinn {
bool fused_supports(const Net&) { return false; }
int64_t fused_workspace_bytes(const Net&, int64_t) { return -1; }
int fused_forward(const Net&, const float*, const float*, int64_t, float*, float*, void*, int64_t, hipStream_t) {
  set_error("fused engine not built"); return PINN_ERR_UNSUPPORTED; }
int fused_loss(const Net&, const LossReq&, const float*, const float*, int64_t, void*, int64_t, hipStream_t) {
  set_error("fused engine not built"); return PINN_ERR_UNSUPPORTED; }
}
