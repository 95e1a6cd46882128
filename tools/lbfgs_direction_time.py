"""Times the pieces of lbfgs._History (batched L-BFGS recursion) on the GPU."""
import torch, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pinn_depthestimation_amd.lbfgs import _History
P, m = 29636, 100
g = torch.randn(P, device="cuda")
h = _History(m, g)
for i in range(m):
    s = torch.randn(P, device="cuda") * 1e-2; y = s * (1 + 0.1 * torch.rand(P, device="cuda"))
    h.push(s, y)
def t(fn, n=200):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("direction      %.1f us" % t(lambda: h.direction(g, 0.5)))
print("push           %.1f us" % t(lambda: h.push(g, g)))
Mk = h.M; b = torch.randn(m, 1, device="cuda", dtype=torch.float64)
print("solve_tri fp64 %.1f us" % t(lambda: torch.linalg.solve_triangular(torch.triu(Mk), b, upper=True)))
Mc = Mk.cpu(); bc = b.cpu()
print("solve_tri cpu  %.1f us" % t(lambda: torch.linalg.solve_triangular(torch.triu(Mc), bc, upper=True)))
print("mv             %.1f us" % t(lambda: torch.mv(h.S, g)))
print("sync roundtrip %.1f us" % t(lambda: torch.stack((g.dot(g), g.abs().sum())).tolist()))
