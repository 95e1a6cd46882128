"""Generate tests/golden/*.npz by running the REFERENCE's own dnn.py / physics.py on CPU.

Run in the build container only (the reference tree does not travel):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_goldens.py

Nothing from the reference is copied: its modules are imported from where they
lie (/root/reference) and only inputs + outputs are written.  The reference has no
tests or golden vectors of its own, so these captures are what pins
the oracle (oracle/pinn_oracle.py) and, through it, the HIP engine.
torch 2.10.0+rocm7.0 CPU kernels, fp32, 8 threads.

G1  state_dict of DNN([3]+[64]*8+[4], 0.0, 'xavier') @ manual_seed(1234); X ~ U(-1,1) (256,3); Y
G2  the 12 compute_gradient(out_c, in_j) columns on G1
G3  Navier_Stokes loss and flat d loss / d theta on G1
G4  physics_equation on DNN([2]+[64]*8+[6]) with the last-layer bias of h set to 0.75 and of
    eta_mean to 0 (conditioning), loss + grad; plus the raw-init 10x10 case
G5  continuity_only / continuity_ftemp on DNN([2]+[20]*4+[3])
G6  loss_func arithmetic of train.py:131-157 with config_CMB.json weights: 12 fidelity points
    + 243 residual points, 2->10x10->6, physics_equation
G7  train.py:188-193 loop, 200 steps of Adam(1e-4)+StepLR(step 50, gamma .8), Navier_Stokes 8x64,
    N = 10000: loss trajectory + final weights
G8  train.py:195-200: ONE torch.optim.LBFGS.step (history 100, strong_wolfe, tol 1e-5/1e-7,
    max_iter 50) from the G7 end state, N = 2000: every closure loss + final weights
G7b the same loop for 1000 steps (StepLR 250 / 0.8), N = 4096: loss trajectory only (`python make_goldens.py g7b`)
G9  train_newmethod.py:120-159 arithmetic on data_at50k.mat's pred_U/pred_V as `trues`
    (2->100x20->3 and 2->8x64->3, continuity_only), seeded synthetic (x,y): 5 Adam steps
"""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import dnn as ref_dnn          # noqa: E402  (reference module, imported in place)
import physics as ref_physics  # noqa: E402

torch.set_num_threads(8)


def sd_np(model):
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def flat_grad(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy().copy()


def cols_of(X, grad_cols):
    """train.py:86-88: each column its own (N,1) tensor; .float() makes it a non-leaf."""
    out = []
    for i in range(X.shape[1]):
        out.append(torch.tensor(X[:, i:i + 1].astype(np.float64), requires_grad=(i in grad_cols)).float())
    return out


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **kw)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in kw.items()})


def g1_g2_g3():
    torch.manual_seed(1234)
    model = ref_dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    rng = np.random.RandomState(1234)
    X = rng.uniform(-1, 1, size=(256, 3)).astype(np.float32)
    c = cols_of(X, (0, 1, 2))
    Y = model(torch.cat(c, dim=-1))
    dY = np.zeros((3, 256, 4), np.float32)
    for j in range(3):
        for o in range(4):
            dY[j, :, o] = ref_physics.compute_gradient(Y[:, o:o + 1], c[j]).detach().numpy()[:, 0]
    h, z, u, v = [Y[:, i:i + 1] for i in range(4)]
    loss = ref_physics.Navier_Stokes(c[0], c[1], c[2], h, z, u, v)
    model.zero_grad()
    loss.backward()
    save("g1_g3_ns_8x64.npz", X=X, Y=Y.detach().numpy(), dY=dY, loss=np.float64(loss.item()),
         grad=flat_grad(model), **{"sd/" + k: v for k, v in sd_np(model).items()})
    return model


def pe_case(layers, N, seed, condition, name):
    torch.manual_seed(seed)
    model = ref_dnn.DNN(layers, 0.0, "xavier")
    if condition:
        with torch.no_grad():
            last = model.layers[-1] if not hasattr(model.layers, f"layer_{len(layers) - 2}") else getattr(
                model.layers, f"layer_{len(layers) - 2}")
            last.bias[0] = 0.75   # h
            last.bias[3] = 0.0    # eta_mean
    rng = np.random.RandomState(seed)
    X = rng.uniform(-1, 1, size=(N, 2)).astype(np.float32)
    out = {}
    for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
        m = ref_dnn.DNN(layers, 0.0, "xavier")
        m.load_state_dict(model.state_dict())
        m = m.to(dt)
        c = [torch.tensor(X[:, i:i + 1].astype(np.float64), requires_grad=True).to(dt) for i in range(2)]
        Y = m(torch.cat(c, dim=-1))
        h, U, V, eta, Hrms, k = [Y[:, i:i + 1] for i in range(6)]
        loss = ref_physics.physics_equation(c[0], c[1], h, U, V, eta, Hrms, k)
        m.zero_grad()
        loss.backward()
        out["loss" + tag] = np.float64(loss.item())
        out["grad" + tag] = flat_grad(m)
        if tag == "":
            out["Y"] = Y.detach().numpy()
    save(name, X=X, **out, **{"sd/" + k: v for k, v in sd_np(model).items()})


def g5():
    torch.manual_seed(55)
    layers = [2] + [20] * 4 + [3]
    model = ref_dnn.DNN(layers, 0.0, "xavier")
    rng = np.random.RandomState(55)
    X = rng.uniform(-1, 1, size=(300, 2)).astype(np.float32)
    X[:, 0] *= 40.0  # so that x < 25.5 (physics.py:26) selects a strict subset
    out = {}
    for fn in ("continuity_only", "continuity_ftemp"):
        c = cols_of(X, (0, 1))
        Y = model(torch.cat(c, dim=-1))
        # train_newmethod.py:136-139,156: outputs ordered trues (U,V) then unknowns (h)
        U, V, h = [Y[:, i:i + 1] for i in range(3)]
        loss = getattr(ref_physics, fn)(c[0], c[1], h, U, V)
        model.zero_grad()
        loss.backward()
        out[fn + "/loss"] = np.float64(loss.item())
        out[fn + "/grad"] = flat_grad(model)
    save("g5_continuity_4x20.npz", X=X, count=np.int64((X[:, 0] < 25.5).sum()), **out,
         **{"sd/" + k: v for k, v in sd_np(model).items()})


def g6():
    cfg = json.load(open(os.path.join(REF, "config_CMB.json")))
    L = cfg["layers"]
    layers = [L["input_features"]] + [L["hidden_width"]] * L["hidden_layers"] + [L["output_features"]]
    torch.manual_seed(66)
    model = ref_dnn.DNN(layers, L["dropout_rate"], L["init_type"])
    with torch.no_grad():
        model.layers.layer_10.bias[0] = 0.75
        model.layers.layer_10.bias[3] = 0.0
    rng = np.random.RandomState(66)
    outs = cfg["data_fidelity"]["outputs"]
    Xf = rng.uniform(-1, 1, size=(cfg["data_fidelity"]["training_points"], 2)).astype(np.float32)
    Tf = np.column_stack([rng.uniform(0.70, 0.80, 12), rng.uniform(-.2, .2, 12), rng.uniform(-.2, .2, 12),
                          rng.uniform(-.05, .05, 12), rng.uniform(.1, .3, 12), rng.uniform(.5, 1.5, 12)]).astype(np.float32)
    Xr = rng.uniform(-1, 1, size=(243, 2)).astype(np.float32)
    # train.py:131-157
    pred = model(torch.tensor(Xf))
    fid = 0
    for i, key in enumerate(outs):
        w = cfg["loss"][f"weight_{key}_loss"]
        fid = fid + w * torch.mean((torch.tensor(Tf[:, i:i + 1]) - pred[:, i:i + 1]) ** 2)
    c = cols_of(Xr, (0, 1))
    Y = model(torch.cat(c, dim=-1))
    named = {key: Y[:, i:i + 1] for i, key in enumerate(cfg["data_residual"]["outputs"])}
    res = ref_physics.physics_equation(c[0], c[1], named["h"], named["U"], named["V"], named["eta_mean"],
                                       named["Hrms"], named["k"])
    loss = cfg["loss"]["weight_fid_loss"] * fid + cfg["loss"]["weight_res_loss"] * res
    model.zero_grad()
    loss.backward()
    save("g6_lossfunc_cmb.npz", Xf=Xf, Tf=Tf, Xr=Xr, fid=np.float64(fid.item()), res=np.float64(res.item()),
         loss=np.float64(loss.item()), grad=flat_grad(model), **{"sd/" + k: v for k, v in sd_np(model).items()})


def g7_g8(model0):
    model = ref_dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict(model0.state_dict())
    rng = np.random.RandomState(77)
    X = rng.uniform(-1, 1, size=(10000, 3)).astype(np.float32)
    c = cols_of(X, (0, 1, 2))

    def loss_func(cc):
        Y = model(torch.cat(cc, dim=-1))
        h, z, u, v = [Y[:, i:i + 1] for i in range(4)]
        return ref_physics.Navier_Stokes(cc[0], cc[1], cc[2], h, z, u, v)

    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.8)
    model.train()
    losses = []
    for i in range(200):               # train.py:188-193
        opt.zero_grad()
        loss = loss_func(c)
        loss.backward()
        opt.step()
        sch.step()
        losses.append(loss.item())
        if i % 50 == 0:
            print("adam", i, losses[-1], flush=True)
    sd_end = sd_np(model)
    save("g7_adam_ns_8x64.npz", X=X, losses=np.array(losses, np.float64),
         **{"sd_end/" + k: v for k, v in sd_end.items()})

    c2 = cols_of(X[:2000], (0, 1, 2))
    lb = torch.optim.LBFGS(model.parameters(), lr=1, max_iter=50, max_eval=None, history_size=100,
                           tolerance_grad=1e-5, tolerance_change=1e-7, line_search_fn="strong_wolfe")
    closure_losses = []

    def closure():                     # train.py:195-199
        lb.zero_grad()
        loss = loss_func(c2)
        loss.backward()
        closure_losses.append(loss.item())
        return loss

    lb.step(closure)
    print("lbfgs evals", len(closure_losses), closure_losses[0], closure_losses[-1])
    save("g8_lbfgs_ns_8x64.npz", losses=np.array(closure_losses, np.float64),
         **{"sd_end/" + k: v for k, v in sd_np(model).items()})


def g9():
    from scipy.io import loadmat
    mat = loadmat(os.path.join(REF, "data_at50k.mat"))
    U, V = mat["pred_U"].astype(np.float32), mat["pred_V"].astype(np.float32)
    N = U.shape[0]
    rng = np.random.RandomState(99)
    X = rng.uniform(-1, 1, size=(N, 2)).astype(np.float32)
    out = {"X": X, "U": U, "V": V}
    for tag, layers in (("100x20", [2] + [20] * 100 + [3]), ("8x64", [2] + [64] * 8 + [3])):
        torch.manual_seed(99)
        model = ref_dnn.DNN(layers, 0.0, "xavier")
        for k, v in sd_np(model).items():
            out[f"{tag}/sd/{k}"] = v
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        sch = torch.optim.lr_scheduler.StepLR(opt, step_size=10000, gamma=0.8)
        c = cols_of(X, (0, 1))
        tU, tV = torch.tensor(U), torch.tensor(V)
        losses = []
        for i in range(5):
            opt.zero_grad()
            pred = model(torch.cat(c, dim=-1))            # train_newmethod.py:123-126
            fid = torch.nn.functional.mse_loss(pred[:, 0:1], tU) + torch.nn.functional.mse_loss(pred[:, 1:2], tV)
            res = ref_physics.continuity_only(c[0], c[1], pred[:, 2:3], pred[:, 0:1], pred[:, 1:2])
            loss = 1 * fid + 1 * res                      # config_CMB_h.json:28-31
            loss.backward()
            if i == 0:
                out[f"{tag}/grad0"] = flat_grad(model)
                out[f"{tag}/fid0"] = np.float64(fid.item())
                out[f"{tag}/res0"] = np.float64(res.item())
            opt.step()
            sch.step()
            losses.append(loss.item())
            print("g9", tag, i, losses[-1], flush=True)
        out[f"{tag}/losses"] = np.array(losses, np.float64)
    save("g9_newmethod_at50k.npz", **out)


if __name__ == "__main__" and len(sys.argv) == 1:
    os.chdir("/tmp")
    m0 = g1_g2_g3()
    pe_case([2] + [64] * 8 + [6], 256, 44, True, "g4_pe_8x64_conditioned.npz")
    pe_case([2] + [10] * 10 + [6], 256, 45, False, "g4_pe_10x10_rawinit.npz")
    g5()
    g6()
    g9()
    g7_g8(m0)


def g7b_long():
    """G7b: 1000 Adam steps (lr 1e-4, StepLR 250/0.8), Navier_Stokes 8x64, N = 4096: how far the
    trajectories stay together.  Run separately: python make_goldens.py g7b"""
    z = np.load(os.path.join(OUT, "g1_g3_ns_8x64.npz"))
    model = ref_dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")})
    rng = np.random.RandomState(707)
    X = rng.uniform(-1, 1, size=(4096, 3)).astype(np.float32)
    c = cols_of(X, (0, 1, 2))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=250, gamma=0.8)
    losses = []
    for i in range(1000):
        opt.zero_grad()
        Y = model(torch.cat(c, dim=-1))
        h, zz, u, v = [Y[:, j:j + 1] for j in range(4)]
        loss = ref_physics.Navier_Stokes(c[0], c[1], c[2], h, zz, u, v)
        loss.backward()
        opt.step()
        sch.step()
        losses.append(loss.item())
        if i % 100 == 0:
            print("g7b", i, losses[-1], flush=True)
    save("g7b_adam_1000_ns_8x64.npz", X=X, losses=np.array(losses, np.float64))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "g7b":
    os.chdir("/tmp")
    g7b_long()
