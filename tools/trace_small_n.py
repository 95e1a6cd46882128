import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from pinn_depthestimation_amd.trainer import PINN
from small_n_latency import ns_config
X = (torch.rand(243, 3, generator=torch.Generator().manual_seed(1234)) * 2 - 1).numpy()
tr = PINN(None, None, X, ns_config(300), log_every=1, checkpoint_every=0)
for _ in range(300): tr.adam_step()
torch.cuda.synchronize()
