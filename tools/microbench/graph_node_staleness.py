#!/usr/bin/env python3
"""Do a captured graph's memset and memcpy nodes still do their work after the device was synchronised and other
kernels ran between replays?  (ROCm 7.2 on MI355X: the memset node does not -- see the output.)
python tools/microbench/graph_node_staleness.py"""
import ctypes

import torch

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
n = 1 << 16
buf = torch.ones(n, device=dev, dtype=torch.int32)      # cleared by a memset node, then incremented by a kernel node
src = torch.arange(n, device=dev, dtype=torch.float32)
dst = torch.zeros(n, device=dev)                        # filled by a memcpy node (contiguous copy_ of the same dtype)
acc = torch.zeros(n, device=dev, dtype=torch.int32)


def body():
    stream = torch.cuda.current_stream().cuda_stream
    assert hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(4 * n), ctypes.c_void_p(stream)) == 0
    buf.add_(1)            # 1 everywhere if the memset ran, more if it did not
    dst.copy_(src)         # hipMemcpyAsync device to device
    acc.add_(buf)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()


def check(tag):
    src.add_(1.0)          # the memcpy node must pick up the new contents
    g.replay()
    torch.cuda.synchronize()
    print("%-44s memset node cleared: %s (buffer max %d)   memcpy node copied: %s" %
          (tag, bool((buf == 1).all()), int(buf.max()), bool(torch.equal(dst, src))), flush=True)


for i in range(3):
    check("replay %d" % i)
for i in range(50):
    g.replay()
check("after 50 replays back to back")
torch.cuda.synchronize()
junk = [torch.full((1 << 18,), 7.0, device=dev) for _ in range(8)]
del junk
check("after a synchronise and foreign kernels")
check("and once more")
