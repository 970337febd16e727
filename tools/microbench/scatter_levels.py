#!/usr/bin/env python3
"""Times nerf_hashgrid_bwd (binned) on level sets of one kind each, to see which kind of level costs what:
python tools/microbench/scatter_levels.py [points]   (run under rocprofv3 --kernel-trace to split bin / reduce)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerf_amd import _lib as L, ops  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.rand(P, 3, generator=g).to(dev)
    for name, res, size in [("2 segments", 22, 10648), ("4 segments", 30, 27000), ("10 segments", 42, 74088),
                            ("24 segments", 58, 195112), ("63 segments", 80, 512000), ("64 segments, hashed", 512, 524288)]:
        n = 8
        d = torch.randn(P, 2 * n, generator=g).to(dev)
        grads = [torch.zeros(size, 2, device=dev) for _ in range(n)]
        work, nbytes = ops.scatter_workspace(P, [size] * n, dev)
        args = (L.ptr(x), P, L.ptr(d), L.int_array([res] * n), L.int_array([size] * n), n, L.ptr_array(grads), L.ptr(work), nbytes,
                L.stream_of(x))
        for _ in range(2):
            L.check(L.lib().nerf_hashgrid_bwd(*args), "bwd")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(L.lib().nerf_hashgrid_bwd(*args), "bwd")
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:22s} {n} levels of res {res:4d} size {size:7d}: {e0.elapsed_time(e1) / 5 / n * 1e3:8.1f} us per level", flush=True)


if __name__ == "__main__":
    main()
