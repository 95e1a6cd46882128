"""TEST INFRASTRUCTURE: a CPU stand-in for pinn_depthestimation_amd.Engine with the two methods bench.py
calls, arithmetic by the oracle (oracle/pinn_oracle.py).  It exists so that the N-rank launcher and the
[grad | sums] all-reduce path of bench.py can be rehearsed with gloo in a container without GPUs
(tests/test_bench_launcher_cpu.py); the product never imports it and bench.py marks its line invalid."""
import torch

from oracle import pinn_oracle as O


class OracleEngine:
    def __init__(self, desc, spec):
        self.desc, self.spec = desc, spec
        torch.set_num_threads(2)

    def residual_loss_grad(self, spec, term_scale, params, X, grad, sums=None):
        assert spec.name == "Navier_Stokes", "the rehearsal evaluator knows the headline residual only"
        p = [q.clone().requires_grad_(True) for q in O.unflatten(params.detach(), self.desc.layers)]
        cols = O.split_columns(X, self.desc.grad_cols)
        Y = O.mlp_forward(p, torch.cat(cols, -1))
        ins = [cols[self.desc.grad_cols[d]] for d in spec.dir_of]
        fields = O.navier_stokes_fields(*ins, *[Y[:, o:o + 1] for o in spec.out_col[:4]])
        terms = torch.stack([(f ** 2).sum() for f in fields])
        grad.add_(O.flat_grad((terms * term_scale).sum(), p))
        if sums is None:
            sums = torch.empty_like(term_scale)
        sums.copy_(terms.detach())
        return sums

    def adam_step(self, params, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        m.lerp_(grad, 1 - beta1)
        v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        denom = (v.sqrt() / ((1 - beta2 ** step) ** 0.5)).add_(eps)
        params.addcdiv_(m, denom, value=-lr / (1 - beta1 ** step))


def make(desc, spec):
    return OracleEngine(desc, spec)
