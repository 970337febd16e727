#!/usr/bin/env python3
"""How long does a replay of the captured hash-grid step take when the GPU was idle before it, or busy with kernels
launched one by one?  Replays back to back, replays each followed by a synchronisation, replays each preceded by a
pause on the host, and replays each preceded by a few small eager kernels on the same stream:
python tools/microbench/graph_replay_gaps.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    early = None
    if os.environ.get("ALLOC_EARLY"):   # the eager kernels' tensors allocated before the step is built and captured
        early = (torch.empty(4096, 64, device=dev), torch.empty(4096, 3, device=dev), torch.empty(4096, 3, device=dev))
    step, _, info = bench.build_ngp_step(0, 4096, dev, None, 1, True)
    assert info.get("hip_graph"), info
    for _ in range(20):
        step()
    torch.cuda.synchronize()

    scratch, src, dst = early if early else (torch.empty(4096, 64, device=dev), torch.empty(4096, 3, device=dev),
                                             torch.empty(4096, 3, device=dev))

    side = torch.cuda.Stream()
    big = torch.empty(1 << 28, device=dev)
    ma = torch.empty(4096, 4096, device=dev, dtype=torch.float16)

    def probe(tag):   # is the card itself slower?  a 1-GiB fill (memory-bound) and a 4096^3 f16 matrix product (clock-bound)
        def t(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 5
        print("probe %-22s fill 1 GiB %.3f ms, 4096^3 f16 matmul %.3f ms" % (tag, t(lambda: big.fill_(1.0)), t(lambda: torch.mm(ma, ma))), flush=True)

    def small():
        for k in range(8):
            if k % 2 == 0:
                scratch.uniform_()
            else:
                dst.copy_(src)

    small_graph = None
    if os.environ.get("SMALL_GRAPH"):   # the same eight small kernels as a second captured graph
        s2 = torch.cuda.Stream()
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            small()
        torch.cuda.current_stream().wait_stream(s2)
        torch.cuda.synchronize()
        small_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(small_graph):
            small()

    def run(n, mode, pause=0.0, eager=0, on_side=False):
        host = 0.0
        ev = []
        t0 = time.time()
        for _ in range(n):
            if pause:
                time.sleep(pause)
            if eager and small_graph is not None:
                small_graph.replay()
            else:
                for k in range(eager):
                    kind = os.environ.get("EAGER_KIND", "mixed")
                    if kind == "uniform" or (kind == "mixed" and k % 2 == 0):
                        scratch.uniform_()
                    elif kind == "fill":
                        scratch.fill_(1.0)
                    elif kind == "nerf":   # one of this library's own kernels through the C ABI
                        from nerf_amd import ops
                        ops.normalize_dirs(src)
                    else:
                        dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h = time.time()
            if on_side:   # the replay on a stream of its own, ordered against the eager work by events
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    e0.record()
                    step()
                    e1.record()
                cur.wait_stream(side)
            else:
                e0.record()
                step()
                e1.record()
            host += time.time() - h
            ev.append((e0, e1))
            if mode == "sync":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        wall = time.time() - t0
        gpu = sum(a.elapsed_time(b) for a, b in ev) / n
        print("%-28s %4d replays: wall %.3f ms each, GPU span of a replay %.3f ms, host time in the call %.3f ms" %
              (mode + (" pause %.1f ms" % (1e3 * pause) if pause else "") + (" %d eager kernels" % eager if eager else "") +
               (" replay on a side stream" if on_side else ""), n, 1e3 * wall / n, gpu, 1e3 * host / n), flush=True)

    if os.environ.get("ALL_ON_SIDE"):   # everything below on a non-default stream
        work = torch.cuda.Stream()
        work.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(work)
    if os.environ.get("QUICK"):
        run(300, "back to back")
        run(300, "back to back", eager=1)
        run(300, "back to back")
        return
    run(500, "back to back")
    for rep in range(1):
        run(500, "back to back")
        run(500, "sync")
        run(300, "sync", pause=0.001)
        run(100, "sync", pause=0.01)
        run(500, "back to back", eager=1)
        run(500, "back to back", eager=8)
        run(500, "sync", eager=8)


if __name__ == "__main__":
    main()
