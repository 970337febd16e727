#!/usr/bin/env python3
"""Per-kernel throughput of the bandwidth-bound stages at data-parallel batch
size (65,536 rays per GPU): algorithmic bytes / HIP-event time, against the
8 TB/s HBM3E peak.  Prints one JSON line per kernel.

    python bench_kernels.py [--rays 65536]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PEAK_HBM_GBS = 8000.0


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def report(name, nbytes, t, note=""):
    gbs = nbytes / t / 1e9
    print(json.dumps({"kernel": name, "ms": round(t * 1e3, 4), "algorithmic_MB": round(nbytes / 1e6, 2),
                      "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / PEAK_HBM_GBS, 4), "note": note}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    args = ap.parse_args()
    from nerf_amd import _lib as L
    from nerf_amd import ops
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.Camera import look_at_camera

    dev = torch.device("cuda:0")
    R, N, NF = args.rays, 64, 128
    NA = N + NF
    g = torch.Generator().manual_seed(0)
    c2w = look_at_camera(torch.tensor([2.0, -3.0, 1.8]), torch.tensor([4.0, -6.0, 3.6]), torch.tensor([0.0, 0.0, 1.0])).to(dev)
    pix = torch.randint(0, 640000, (R,), generator=g, dtype=torch.int32).to(dev)
    u = torch.rand(R, N, generator=g).to(dev)
    uf = torch.rand(R, NF, generator=g).to(dev)
    z_lin = torch.linspace(2.0, 6.0, N, device=dev)
    f = 1111.11

    # outputs are allocated once and the C entry points called directly: at 20-50 us per kernel the
    # Python wrappers (allocation, autograd bookkeeping) would be what is timed
    # what this chip sustains for plain streams of the same sizes (torch's fill and copy kernels): the practical
    # ceilings next to the nominal 8 TB/s
    for mb in (86, 304, 1107):
        n = mb * 250000
        a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
        t = timeit(lambda: a.fill_(1.0))
        report("reference: fill %d MB (write only)" % mb, 4 * n, t, "torch fill_")
        t = timeit(lambda: b.copy_(a))
        report("reference: copy %d MB (read + write)" % mb, 8 * n, t, "torch copy_")
        del a, b
    lib = L.lib()
    st = L.stream_of(u)
    o, d, z, pts = ops.raygen_sample(c2w, f, f, 400.0, 400.0, 800, 800, pix, z_lin, u)
    c2w_c = c2w.contiguous()
    a_rs = (L.ptr(c2w_c), f, f, 400.0, 400.0, 800, 800, L.ptr(pix), R, L.ptr(z_lin), L.ptr(u), N, L.ptr(o), L.ptr(d), L.ptr(z),
            L.ptr(pts), st)   # arguments marshalled once: the loop below should time the kernel, not Python
    t = timeit(lambda: lib.nerf_raygen_sample(*a_rs))
    report("raygen_sample (64 samples)", R * (4 + 24) + R * N * (4 + 4 + 12), t, "read pix,u; write rays,z,pts")

    for n, zz in ((N, z), (NA, torch.sort(torch.rand(R, NA, generator=g) * 4 + 2, dim=-1).values.to(dev))):
        rgb = torch.rand(R, n, 3, generator=g).to(dev)
        sigma = (torch.rand(R, n, generator=g) * 0.3).to(dev)
        out = torch.empty(R, 3, device=dev)
        w = torch.empty(R, n, device=dev)
        dep = torch.empty(R, device=dev)
        lib = L.lib()
        st = L.stream_of(rgb)
        a_cf = (L.ptr(rgb), L.ptr(sigma), L.ptr(zz), R, n, 1, 1e5, L.ptr(out), L.ptr(w), L.ptr(dep), st)
        t = timeit(lambda: lib.nerf_composite_fwd(*a_cf))
        report("composite_fwd (%d samples)" % n, R * n * 24 + R * 16, t, "read rgb,sigma,z; write weights,colour,depth")
        d_out = torch.randn(R, 3, generator=g).to(dev)
        d_rgb, d_sig = torch.empty_like(rgb), torch.empty_like(sigma)
        a_cb = (L.ptr(rgb), L.ptr(sigma), L.ptr(zz), L.ptr(d_out), None, None, R, n, 1, 1e5, L.ptr(d_rgb), L.ptr(d_sig), st)
        t = timeit(lambda: lib.nerf_composite_bwd(*a_cb))
        report("composite_bwd (%d samples)" % n, R * n * 36 + R * 12, t, "read rgb,sigma,z,d_out; write d_rgb,d_sigma")
        if n == N:
            wts = w.clone()

    z_fine, idx, z_all, pts_all = ops.pdf_sample(o, d, z, wts, uf)
    a_ps = (L.ptr(o), L.ptr(d), L.ptr(z), L.ptr(wts), L.ptr(uf), R, N, NF, L.ptr(z_fine), L.ptr(idx), L.ptr(z_all), L.ptr(pts_all), st)
    t = timeit(lambda: lib.nerf_pdf_sample(*a_ps))
    report("pdf_sample (64 -> +128)", R * (N * 8 + NF * 4 + 24) + R * NA * 16 + R * NF * 4, t,
           "read z,weights,u,rays; write z_fine,z_all,pts")
    uf_sorted = torch.sort(uf, dim=-1).values
    a_ps2 = a_ps[:4] + (L.ptr(uf_sorted),) + a_ps[5:]
    t = timeit(lambda: lib.nerf_pdf_sample(*a_ps2))
    report("pdf_sample (64 -> +128), uniforms sorted", R * (N * 8 + NF * 4 + 24) + R * NA * 16 + R * NF * 4, t,
           "the same with sorted uniforms (deterministic renders): the kernel's sort is skipped")

    x = pts.reshape(-1, 3)
    P = x.shape[0]
    t = timeit(lambda: ops.encode_pe(x, 10))
    report("encode_pe (L=10, stand-alone)", P * (12 + 252), t, "the fused MLP never materialises this")

    m = FastNGPNeRF().to(dev)
    xs = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    tabs = [e.weight.detach() for e in m.hash_encoder.embeddings]
    res = m.hash_encoder.resolutions
    t = timeit(lambda: ops.hashgrid_debug(xs, tabs, res)[0] if False else ops.hashgrid_encode(xs, tabs, res))
    report("hashgrid_fwd (16 levels)", P * (16 * 8 * 8 + 12 + 128), t, "1,024 B/point of 8-byte gathers")
    d_out = torch.randn(P, 32, generator=g).to(dev)
    grads = [torch.zeros_like(tb) for tb in tabs]
    sizes = [int(tb.shape[0]) for tb in tabs]
    lib = L.lib()
    t = timeit(lambda: L.check(lib.nerf_hashgrid_bwd(L.ptr(xs), P, L.ptr(d_out), L.int_array(res), L.int_array(sizes), 16,
                                                     L.ptr_array(grads), None, 0, L.stream_of(xs)), "hbwd"), iters=5)
    report("hashgrid_bwd (16 levels, direct atomics)", P * (16 * 8 * 8 + 12 + 128), t, "1,024 B/point of float atomics")
    work, nbytes = ops.scatter_workspace(P, sizes, dev)
    t = timeit(lambda: L.check(lib.nerf_hashgrid_bwd(L.ptr(xs), P, L.ptr(d_out), L.int_array(res), L.int_array(sizes), 16,
                                                     L.ptr_array(grads), L.ptr(work), nbytes, L.stream_of(xs)), "hbwd"), iters=5)
    report("hashgrid_bwd (16 levels, binned by table segment)", P * (16 * 8 * 8 + 12 + 128), t,
           "updates sorted by segment through a %d MB workspace, summed in LDS" % (nbytes >> 20))

    dirs = torch.randn(R, 3, generator=g).to(dev)
    with torch.no_grad():
        t = timeit(lambda: m(xs, dirs, dirs_div=N), iters=5)
    report("ngp_fwd fused (hash + MLPs, inference)", P * (16 * 8 * 8 + 12 + 16), t, "gather-bound; 13.8 kFLOP/point of MFMA on top")

    # the vanilla MLP's inference forward (what render_image runs): matrix-core work, reported as FLOP/s
    from nerf_amd.models.NeRF import NeRF
    vm = NeRF().to(dev)
    with torch.no_grad():
        t = timeit(lambda: ops.vanilla_mlp(xs, dirs, vm._params(), False, dirs_div=N), iters=5)
    flop = 2 * 593408 * P
    print(json.dumps({"kernel": "vanilla_fwd (8x256 MLP, inference, %d points)" % P, "ms": t * 1e3,
                      "fp32_equivalent_TFLOP/s": flop / t / 1e12, "frac_of_f16_peak_over_3": flop / t / 1e12 / (2500.0 / 3),
                      "Mpoints/s": P / t / 1e6, "note": "no activations saved; 3 f16 MFMAs per fp32 product"}), flush=True)


if __name__ == "__main__":
    main()
