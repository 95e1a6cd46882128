"""Round-2 goldens, generated like make_goldens.py by running the REFERENCE's dnn.py / physics.py in
place on CPU (build container only; nothing of the reference is copied, only inputs + outputs are kept):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_goldens_r2.py [g10|g10b|g8b|g8s|g9x]

G10  BASELINE configs[3] as configured: reference DNN([3]+[256]*12+[4]) + physics.Navier_Stokes, N = 2000,
     fp32 AND fp64 loss / flat gradient (weights from tests/golden/synth.py, seed in the fixture).
G10b 100 Adam(1e-4) steps of the same problem on the reference in fp32: loss trajectory.
G8b  SURVEY §8c G8 second half: scipy.optimize.minimize(L-BFGS-B, jac=True, maxcor 50, maxls 50) for 50
     iterations over a flat float64 closure built from the reference's dnn.DNN / physics.Navier_Stokes
     (fp32 network, as the stale l_bfgs_b_optimizer wrapper did); start = G7 end state, N = 2000.
G8s  the reference's own torch.optim.LBFGS trajectory (G8) re-run at 1 thread: its thread-count spread
     is the noise floor a tolerance on G8 has to respect.
G9x  fp64 re-evaluation of G9's first-iteration fidelity / residual / gradient (100x20 and 8x64 nets).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_goldens as MG          # noqa: E402  (helpers; imports the reference modules in place)
import synth                       # noqa: E402

ref_dnn, ref_physics = MG.ref_dnn, MG.ref_physics
OUT = HERE


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def ns_loss(model, c):
    Y = model(torch.cat(c, dim=-1))
    h, z, u, v = [Y[:, i:i + 1] for i in range(4)]
    return ref_physics.Navier_Stokes(c[0], c[1], c[2], h, z, u, v), Y


G10_LAYERS = [3] + [256] * 12 + [4]
G10_SEED = 1010


def g10_model(dtype=torch.float32):
    m = ref_dnn.DNN(G10_LAYERS, 0.0, "xavier")
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_of(synth.xavier_params(G10_LAYERS, G10_SEED)).items()}
    m.load_state_dict(sd)
    return m.to(dtype)


def g10():
    rng = np.random.RandomState(G10_SEED + 1)
    X = rng.uniform(-1, 1, size=(2000, 3)).astype(np.float32)
    out = {}
    for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
        m = g10_model(dt)
        c = [torch.tensor(X[:, i:i + 1].astype(np.float64), requires_grad=True).to(dt) for i in range(3)]
        loss, Y = ns_loss(m, c)
        m.zero_grad()
        loss.backward()
        out["loss" + tag] = np.float64(loss.item())
        out["grad" + tag] = MG.flat_grad(m)
        out["Y" + tag] = Y.detach().numpy()
    # the reference's own fp32-vs-fp64 disagreement at this shape (the noise floor of any fp32 engine)
    ref_grad_err = rel_l2(out["grad32"], out["grad64"])
    ref_loss_err = abs(out["loss32"] - out["loss64"]) / abs(out["loss64"])
    print("G10: loss32", out["loss32"], "loss64", out["loss64"], "ref fp32-vs-fp64: loss", ref_loss_err, "grad", ref_grad_err)
    MG.save("g10_ns_12x256.npz", X=X, seed=np.int64(G10_SEED), loss32=out["loss32"], loss64=out["loss64"],
            grad64=out["grad64"].astype(np.float32),       # fp64 gradient stored at fp32 (rounding 6e-8 << tolerances)
            Y32=out["Y32"], ref_loss_err=np.float64(ref_loss_err), ref_grad_err=np.float64(ref_grad_err))


def g10b():
    z = np.load(os.path.join(OUT, "g10_ns_12x256.npz"))
    X = z["X"]
    m = g10_model()
    c = MG.cols_of(X, (0, 1, 2))
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=10000, gamma=0.8)
    m.train()
    losses = []
    for i in range(100):
        opt.zero_grad()
        loss, _ = ns_loss(m, c)
        loss.backward()
        opt.step()
        sch.step()
        losses.append(loss.item())
        if i % 10 == 0:
            print("g10b", i, losses[-1], flush=True)
    MG.save("g10b_adam_ns_12x256.npz", losses=np.array(losses, np.float64))


def g7_end_model():
    z = np.load(os.path.join(OUT, "g7_adam_ns_8x64.npz"))
    m = ref_dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    m.load_state_dict({k[len("sd_end/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_end/")})
    return m, z["X"]


def g8b(threads=8, save=True):
    from scipy.optimize import minimize
    torch.set_num_threads(threads)
    m, X = g7_end_model()
    c = MG.cols_of(X[:2000], (0, 1, 2))
    params = list(m.parameters())
    sizes = [p.numel() for p in params]

    def set_flat(x):
        off = 0
        with torch.no_grad():
            for p, n in zip(params, sizes):
                p.copy_(torch.from_numpy(x[off:off + n].astype(np.float32)).view_as(p))
                off += n

    evals = []

    def fun(x):                               # flat float64 -> (loss, flat float64 grad), fp32 network
        set_flat(x)
        m.zero_grad()
        loss, _ = ns_loss(m, c)
        loss.backward()
        evals.append(loss.item())
        return float(loss.item()), MG.flat_grad(m).astype(np.float64)

    x0 = np.concatenate([p.detach().numpy().reshape(-1) for p in params]).astype(np.float64)
    acc = []

    def cb2(intermediate_result):
        acc.append(float(intermediate_result.fun))

    res = minimize(fun, x0, jac=True, method="L-BFGS-B", callback=cb2,
                   options={"maxiter": 50, "maxfun": 50000, "maxcor": 50, "maxls": 50,
                            "ftol": 1.0 * np.finfo(float).eps})
    print("g8b: threads", threads, "nit", res.nit, "nfev", res.nfev, "f0", evals[0], "f_end", res.fun, res.message)
    torch.set_num_threads(8)
    if not save:
        return np.array(evals, np.float64), float(res.fun)
    MG.save("g8b_scipy_lbfgsb_ns_8x64.npz", evals=np.array(evals, np.float64), accepted=np.array(acc, np.float64),
            nit=np.int64(res.nit), nfev=np.int64(res.nfev), fun=np.float64(res.fun),
            x_end=res.x.astype(np.float32))


def g8s():
    """G8 (torch.optim.LBFGS) and G8b (SciPy L-BFGS-B) again at 1, 2 and 4 threads: how far the REFERENCE's own
    trajectories move when only the summation order of its fp32 kernels changes.  Stored: the per-evaluation maximum
    over the three re-runs of |loss(k threads) - loss(8 threads)| / loss(8 threads), and the end-loss spread."""
    z8 = np.load(os.path.join(OUT, "g8_lbfgs_ns_8x64.npz"))
    zb = np.load(os.path.join(OUT, "g8b_scipy_lbfgsb_ns_8x64.npz"))
    sp_t, sp_s, end_s = None, None, []
    for th in (1, 2, 4):
        torch.set_num_threads(th)
        m, X = g7_end_model()
        c2 = MG.cols_of(X[:2000], (0, 1, 2))
        lb = torch.optim.LBFGS(m.parameters(), lr=1, max_iter=50, max_eval=None, history_size=100,
                               tolerance_grad=1e-5, tolerance_change=1e-7, line_search_fn="strong_wolfe")
        losses = []

        def closure():
            lb.zero_grad()
            loss, _ = ns_loss(m, c2)
            loss.backward()
            losses.append(loss.item())
            return loss

        lb.step(closure)
        torch.set_num_threads(8)
        a, b = np.array(losses), z8["losses"]
        n = min(len(a), len(b))
        s1 = np.full(len(b), np.nan); s1[:n] = np.abs(a[:n] - b[:n]) / np.abs(b[:n])
        sp_t = s1 if sp_t is None else np.fmax(sp_t, s1)
        ev, fend = g8b(th, save=False)
        n = min(len(ev), len(zb["evals"]))
        s2 = np.full(len(zb["evals"]), np.nan); s2[:n] = np.abs(ev[:n] - zb["evals"][:n]) / zb["evals"][:n]
        sp_s = s2 if sp_s is None else np.fmax(sp_s, s2)
        end_s.append(abs(fend - float(zb["fun"])) / float(zb["fun"]))
        print("g8s threads", th, "torch first10 %.2e first20 %.2e all %.2e | scipy first12 %.2e all %.2e end %.2e"
              % (np.nanmax(s1[:10]), np.nanmax(s1[:20]), np.nanmax(s1), np.nanmax(s2[:12]), np.nanmax(s2), end_s[-1]))
    MG.save("g8s_lbfgs_thread_spread.npz", spread=sp_t, scipy_spread=sp_s, scipy_end_spread=np.float64(max(end_s)))


def g9x():
    z = np.load(os.path.join(OUT, "g9_newmethod_at50k.npz"))
    X, U, V = z["X"], z["U"], z["V"]
    out = {}
    for tag, layers in (("100x20", [2] + [20] * 100 + [3]), ("8x64", [2] + [64] * 8 + [3])):
        m = ref_dnn.DNN(layers, 0.0, "xavier")
        pre = f"{tag}/sd/"
        m.load_state_dict({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
        m = m.double()
        c = [torch.tensor(X[:, i:i + 1].astype(np.float64), requires_grad=True) for i in range(2)]
        pred = m(torch.cat(c, dim=-1))
        tU, tV = torch.tensor(U.astype(np.float64)), torch.tensor(V.astype(np.float64))
        fid = torch.nn.functional.mse_loss(pred[:, 0:1], tU) + torch.nn.functional.mse_loss(pred[:, 1:2], tV)
        res = ref_physics.continuity_only(c[0], c[1], pred[:, 2:3], pred[:, 0:1], pred[:, 1:2])
        loss = fid + res
        m.zero_grad()
        loss.backward()
        g64 = MG.flat_grad(m)
        out[f"{tag}/fid64"] = np.float64(fid.item())
        out[f"{tag}/res64"] = np.float64(res.item())
        out[f"{tag}/grad64"] = g64.astype(np.float32)
        out[f"{tag}/ref_grad_err"] = np.float64(rel_l2(z[f"{tag}/grad0"], g64))
        out[f"{tag}/ref_res_err"] = np.float64(abs(float(z[f"{tag}/res0"]) - res.item()) / abs(res.item()))
        out[f"{tag}/ref_fid_err"] = np.float64(abs(float(z[f"{tag}/fid0"]) - fid.item()) / abs(fid.item()))
        print("g9x", tag, "ref fp32-vs-fp64: fid", out[f"{tag}/ref_fid_err"], "res", out[f"{tag}/ref_res_err"],
              "grad", out[f"{tag}/ref_grad_err"])
    MG.save("g9x_newmethod_fp64.npz", **out)


if __name__ == "__main__":
    os.chdir("/tmp")
    which = sys.argv[1:] or ["g10", "g10b", "g8b", "g8s", "g9x"]
    for w in which:
        {"g10": g10, "g10b": g10b, "g8b": g8b, "g8s": g8s, "g9x": g9x}[w]()
