"""A training (or rendering) step captured once into a HIP graph and replayed.

A step of the hash-grid model at 4096 rays is some 25 kernel launches of 5-200 us each: issued one by one from
Python the GPU waits for the host between them (a fifth of the step).  Everything the step does goes through
``torch``'s allocator and the current stream -- the C entry points take the stream as an argument and allocate
nothing -- so ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) can record the whole of it, backward and optimiser
included, and replay it as one submission.

Requirements on the step: it reads its inputs from tensors that stay where they are (new data is copied INTO
them), draws random numbers with ``torch.rand(..., device=...)`` only, does nothing that waits for the GPU
(``.item()``, ``.cpu()``, printing a tensor), no host-to-device copies, no collectives, and its optimiser keeps the
step count on the device (``FusedAdam(capturable=True)``).  The reference has no counterpart (its loops are eager
PyTorch); a captured step computes what the eager step computes.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    """``g = GraphedStep(step, warmup=3)`` runs ``step()`` eagerly ``warmup`` times on a side stream (lazy
    initialisation: optimiser state, kernel attributes, allocator pools), captures one more call, and from then
    on ``g()`` replays it and returns the tensors the captured call returned (the same objects every time,
    holding the latest values).  The warm-up and captured calls are real steps: they update the model."""

    def __init__(self, step: Callable[[], object], warmup: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = step()
        self.steps_run = max(1, warmup)   # eager steps taken before the first replay (the capture itself does not execute)

    def __call__(self):
        self.graph.replay()
        return self.out
