"""Ray-sharded data parallelism: one process per GPU, ``torch.distributed`` with
the ``nccl`` backend (RCCL over xGMI on ROCm); ``gloo`` works for CPU rehearsal.

The reference is single-process.  Rays are independent, so the only exchange of
a training step is the parameter-gradient average: every rank renders its own
ray batch with replicated parameters, the gradients are summed in ONE flat
all-reduce per step (2.4 MB per vanilla MLP, 41.8 MB for the hash-grid model)
and every rank applies the same optimiser update.  Inference needs only a final
gather of the rendered rows.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of n items for this rank (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class GradAverager:
    """Averages ``p.grad`` of the given parameters over all ranks with a single
    flat all-reduce.  Parameters whose grad is None on this rank contribute zeros."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def average(self) -> None:
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
        have = [(p.grad, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        if have:
            torch._foreach_copy_([v for _, v in have], [g for g, _ in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def gather_rows(local: torch.Tensor, sizes: List[int], group=None) -> torch.Tensor:
    """Concatenate per-rank row blocks [n_r, ...] (n_r = sizes[r]) on every rank:
    the image gather that closes a sharded inference render."""
    world = dist.get_world_size(group)
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
