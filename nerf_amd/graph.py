"""A training (or rendering) step captured once into a HIP graph and replayed.

A step of the hash-grid model at 4096 rays is some 25 kernel launches of 5-200 us each: issued one by one from
Python the GPU waits for the host between them (a fifth of the step).  Everything the step does goes through
``torch``'s allocator and the current stream -- the C entry points take the stream as an argument and allocate
nothing -- so ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) can record the whole of it, backward and optimiser
included, and replay it as one submission.

Requirements on the step: it reads its inputs from tensors that stay where they are (new data is copied INTO
them), draws random numbers with ``torch.rand(..., device=...)`` only, does nothing that waits for the GPU
(``.item()``, ``.cpu()``, printing a tensor), no host-to-device copies, no collectives, and its optimiser keeps the
step count and the learning rate on the device (``FusedAdam(capturable=True)``).  The reference has no counterpart
(its loops are eager PyTorch); a captured step computes what the eager step computes -- including under a learning-rate
scheduler (src/train_nerf.py:87,204; src/train_nerf_ngp.py:57-62), whose rate reaches the device in front of each replay.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    """``g = GraphedStep(step, warmup=3)`` runs ``step()`` eagerly ``warmup`` times on a side stream (lazy
    initialisation: optimiser state, kernel attributes, allocator pools), captures one more call, and from then
    on ``g()`` replays it and returns the tensors the captured call returned (the same objects every time,
    holding the latest values).  Only the warm-up calls are real steps that update the model: the captured call is
    recorded, not executed (``steps_run`` counts the steps actually taken) -- its host-side effects still happen once,
    which is why an optimiser that counts its steps on the host cannot be part of a captured step.

    ``optimizers``: the optimisers ``step`` calls.  Each must keep its step state on the device (every parameter group
    ``capturable``; anything else raises) and is asked to ``sync_lr()`` in front of every replay, so that
    ``scheduler.step()`` between replays takes effect exactly as between eager steps.
    ``prepare(i)``: called in front of warm-up step ``i`` (to copy that step's data into the input tensors; in front of a
    replay the caller does that itself)."""

    def __init__(self, step: Callable[[], object], warmup: int = 3, optimizers=(), prepare=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        # The graph holds the ADDRESSES of everything the step read: the step function is kept, and with it the tensors
        # its closure owns.  (Dropped, they go back to the allocator, whose next client overwrites what the replays read.)
        self.step = step
        self.optimizers = list(optimizers)
        for opt in self.optimizers:
            if not all(g.get("capturable", False) for g in opt.param_groups):
                raise ValueError("GraphedStep: every optimiser of a captured step must be capturable "
                                 "(FusedAdam(capturable=True)): a host-side step count would be frozen into the graph")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(max(1, warmup)):
                if prepare is not None:
                    prepare(i)
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for opt in self.optimizers:
            opt.sync_lr()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = step()
        self.steps_run = max(1, warmup)   # eager steps taken before the first replay (the capture itself does not execute)

    def __call__(self):
        for opt in self.optimizers:
            opt.sync_lr()
        self.graph.replay()
        return self.out
