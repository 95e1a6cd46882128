"""parallel — data parallelism over collocation points.

Every loss term of the reference is a mean over points (physics.py:24,28,45,86,118;
train.py:141), so with points sharded across ranks
    loss = sum_r S_r / N_global,   grad = sum_r G_r / N_global.
Each rank runs the fused kernel on its shard with term_scale = weight / N_global and ONE
all-reduce (sum) of the flat fp32 buffer [grad (P) | loss sums] follows per closure
evaluation — 116 KiB for 8x64, latency-bound on xGMI.  Parameters and optimiser state are
replicated; identical arithmetic on identical all-reduced gradients keeps them identical, so
parameters are broadcast once at start and never again.  One process per GPU;
torch.distributed backend "nccl" is RCCL on ROCm ("gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal slices; the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class Reducer:
    """Sum-all-reduce over the data-parallel group; a no-op for a single process."""

    def __init__(self, group=None, force: bool = False):
        """force = True runs every collective even in a world of one (rehearsal of the RCCL path on a single GPU)."""
        self.group = group
        self.force = bool(force)
        self.dist = None
        self.world, self.rank = 1, 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.world = dist.get_world_size(group)
                self.rank = dist.get_rank(group)
        except Exception:          # pragma: no cover
            pass

    @property
    def active(self) -> bool:
        return self.world > 1 or (self.force and self.dist is not None)

    def allreduce_sum_(self, buf: torch.Tensor) -> torch.Tensor:
        if self.active:
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
        return buf

    def broadcast_(self, buf: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.active:
            self.dist.broadcast(buf, src=src, group=self.group)
        return buf

    def shard(self, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """This rank's contiguous slice of the rows of t."""
        if t is None or not self.active:
            return t
        lo, hi = shard_bounds(t.shape[0], self.rank, self.world)
        return t[lo:hi]

    def gather_rows(self, local: torch.Tensor, n_total: int) -> Optional[torch.Tensor]:
        """Inverse of shard(): the (n_total, ...) tensor whose contiguous row slices the ranks hold, on
        rank 0 (None elsewhere).  Shards differ by at most one row, so every rank pads to the largest
        and ONE all_gather moves the data (collective: every rank must call it)."""
        if not self.active:
            return local
        rows = [shard_bounds(n_total, r, self.world) for r in range(self.world)]
        most = max(hi - lo for lo, hi in rows)
        pad = torch.zeros((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(parts, pad, group=self.group)
        if self.rank != 0:
            return None
        return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, rows)], 0)
