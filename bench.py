#!/usr/bin/env python3
"""Headline benchmark: rays/sec of one NeRF training step (forward + backward +
Adam) on synthetic Blender-lego-shaped ray batches, 64 coarse + 128 fine
samples, vanilla 8x256 MLPs (coarse and fine), fp32.

    python bench.py --gpus N --steps K --warmup W

One GPU: the 4096-ray batch the metric is quoted on.  N > 1 is launched by
torch.distributed.run, one rank per GPU: every rank renders its own 65,536-ray
batch in ray chunks (rays are independent, weak scaling) and the parameter
gradients are summed with one RCCL all-reduce per model, issued under the last
chunk's backward.  Rank 0 prints ONE JSON line; the hash-grid configuration
rides along in it as "secondary".
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
FOCAL = 0.5 * 800 / math.tan(0.5 * 0.6911112)   # Blender lego camera_angle_x
NEAR, FAR = 2.0, 6.0
N_COARSE, N_FINE = 64, 128
# Algorithmic FLOPs per point evaluation of the 8x256 MLP (593,408 MACs forward):
FLOP_FWD = 2 * 593408
FLOP_BWD_DATA = 2 * (593408 - 63 * 256 - 63 * 256 - 27 * 128)   # no gradient into the encodings
FLOP_WGRAD = 2 * 593408
PEAK_F32_MATRIX_TFLOPS = 157.3
# All three MLP kernels form an fp32 product from 3 f16 MFMAs (operands split in two f16 pieces, the three
# partial products that matter): the matrix roofline of an fp32-equivalent FLOP is the dense f16 peak / 3.
PEAK_F16_MATRIX_TFLOPS = 2500.0
MFMA_PER_PRODUCT = 3
PEAK_HBM_GBS = 8000.0
# Algorithmic HBM bytes per point of the three kernels (every array read or written exactly once):
# forward: x, rgb, sigma; the saved activations x_enc 64, d_enc 32, h0..h7 8x256, feat 256, hr 128 floats; 288 B of ReLU bits
BYTES_FWD = 4 * (3 + 3 + 1) + 4 * (64 + 32 + 8 * 256 + 256 + 128) + 288
# backward-data: d y0..7, d feat, d hr, 4 head gradients written; ReLU bits, d_rgb, d_sigma, rgb, sigma read
BYTES_BWD_DATA = 4 * (8 * 256 + 256 + 128 + 4) + 288 + 4 * (3 + 1 + 3 + 1)
# weight gradients: every saved activation and every gradient matrix read once
BYTES_WGRAD = 4 * (8 * 256 + 256 + 128) + 4 * (64 + 32 + 8 * 256 + 256)


def lego_pose(seed: int) -> torch.Tensor:
    """Look-at pose on the radius-4.03 sphere of the Blender scenes (reference camera convention)."""
    from nerf_amd.Camera import look_at_camera
    g = torch.Generator().manual_seed(seed)
    th = float(torch.rand(1, generator=g)) * 2 * math.pi
    ph = math.radians(30.0)
    eye = 4.03 * torch.tensor([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph), math.sin(ph)])
    # the reference's rays leave along -forward: aim the look-at at the mirrored point
    return look_at_camera(eye, 2 * eye, torch.tensor([0.0, 0.0, 1.0]))


def make_batch(rank: int, rays: int, dev):
    g = torch.Generator().manual_seed(1 + rank)
    pix = torch.randperm(H * W, generator=g)[:rays].to(torch.int32)
    g2 = torch.Generator().manual_seed(2 + 1000 * rank)
    u_c = torch.rand(rays, N_COARSE, generator=g2)
    u_f = torch.rand(rays, N_FINE, generator=g2)
    g3 = torch.Generator().manual_seed(3 + 1000 * rank)
    gt = torch.rand(rays, 3, generator=g3)
    return lego_pose(0).to(dev), pix.to(dev), u_c.to(dev), u_f.to(dev), gt.to(dev)


NGP_NEAR, NGP_FAR = 0.05, 1.2   # src/train_nerf_ngp.py:32-33 (unit-sphere-normalised poses)
RAYS_SINGLE, RAYS_DP = 4096, 65536
OVERLAP = os.environ.get("NERF_DP_OVERLAP", "1") != "0"   # all-reduce each bucket from its gradient hooks (under backward)
NGP_WORKLOAD = ("Instant-NGP variant (16 levels, 2^19 rows, 64-wide MLPs), 800x800 camera on the unit sphere, 64 samples, "
                "%d-ray batch per GPU, one training step incl. Adam over the tables (config.hip_graph true: the step is "
                "captured once into a HIP graph and replayed; false: issued launch by launch)")


def ball_grid(radius, dev, res=32):
    """uint8 occupancy grid [res^3] (indexed x + 32 y + 1024 z like the reference's xyz_to_idx): the cells whose centre
    lies in a ball around the origin -- a synthetic stand-in for the grid get_density_grid derives from a trained model
    (src/utils/analyze_capture_results.py:233-248).  Returns (grid, fraction of the cells that are occupied)."""
    c = (torch.arange(res).float() + 0.5) / res * 2 - 1
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    occ = (x ** 2 + y ** 2 + z ** 2) < radius ** 2
    ix, iy, iz = torch.meshgrid(torch.arange(res), torch.arange(res), torch.arange(res), indexing="ij")
    grid = torch.zeros(res ** 3, dtype=torch.uint8)
    grid[(ix + iy * 32 + iz * 1024).reshape(-1)] = occ.reshape(-1).to(torch.uint8)
    return grid.to(dev), float(occ.float().mean())


def grid_keeping(fraction, pts, dev):
    """The ball grid that keeps about `fraction` of the samples `pts` (bisection on the radius, at set-up)."""
    from nerf_amd import ops
    lo, hi = 0.0, 2.0
    for _ in range(14):
        mid = 0.5 * (lo + hi)
        grid, _ = ball_grid(mid, dev)
        kept = ops.occupancy_compact(pts, grid, 32)[0].numel() / float(pts.shape[0] * pts.shape[1])
        lo, hi = (mid, hi) if kept < fraction else (lo, mid)
    grid, cells = ball_grid(hi, dev)
    kept = ops.occupancy_compact(pts, grid, 32)[0].numel() / float(pts.shape[0] * pts.shape[1])
    return grid, {"grid_cells_occupied": cells, "samples_kept": kept, "ball_radius": hi}


def build_step(kind, rank, rays, dev, chunk_rays, capturable=False, occupancy=None, skip_mode=True):
    """One training step of the vanilla (coarse + fine 8x256 MLPs, 64 + 128 samples) or hash-grid workload:
    fused ray generation + stratified sampling, chunked forward/backward with the gradients in flat buckets
    (their cross-rank sum starts under the last chunk's backward), fused Adam.  Returns (step, model)."""
    from nerf_amd import ops
    from nerf_amd.dist import GradBuckets
    from nerf_amd.optim import FusedAdam
    from nerf_amd.render import nerf_loss, ngp_loss, train_step   # ngp_loss: MSE + the weight regulariser of src/train_nerf_ngp.py:221-222

    c2w, pix, u_c, u_f, gt = make_batch(rank, rays, dev)
    if kind == "ngp":
        # the hash-grid variant: 64 samples, one model, the loss and optimiser settings of
        # src/train_nerf_ngp.py:221-222 and :55; camera on the unit sphere, near/far of :32-33
        from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
        torch.manual_seed(0)
        model = FastNGPNeRF().to(dev)
        buckets = GradBuckets([list(model.hash_encoder.parameters()),
                               [p for n, p in model.named_parameters() if not n.startswith("hash_encoder")]], overlap=OVERLAP)
        opt = FusedAdam(buckets.parameters(), lr=1e-2, eps=1e-15, capturable=capturable)
        c2w = c2w.clone()
        c2w[:, 3] = c2w[:, 3] / 4.03
        z_lin = torch.linspace(NGP_NEAR, NGP_FAR, N_COARSE).to(dev)
        kw = dict(loss_fn=ngp_loss, white_bkgd=False)
        near, far = NGP_NEAR, NGP_FAR
        if occupancy:
            pts0 = ops.raygen_sample(c2w, FOCAL, FOCAL, W * 0.5, H * 0.5, H, W, pix, z_lin, u_c)[3]
            grid, occ_info = grid_keeping(occupancy, pts0, dev)
            kw.update(density_grid=grid, grid_res=32, skip_empty=skip_mode)
    else:
        from nerf_amd.models.NeRF import NeRF
        torch.manual_seed(0)
        model = NeRF().to(dev)
        torch.manual_seed(1)
        fine = NeRF().to(dev)
        # backward reaches the fine model first: its bucket's all-reduce runs under the coarse model's backward
        buckets = GradBuckets([list(fine.parameters()), list(model.parameters())], overlap=OVERLAP)
        opt = FusedAdam(buckets.parameters(), lr=5e-4)   # torch.optim.Adam's update in one launch (tests/test_gpu_optim.py)
        z_lin = torch.linspace(NEAR, FAR, N_COARSE).to(dev)
        kw = dict(loss_fn=nerf_loss, white_bkgd=True, fine_model=fine, n_importance=N_FINE, u_fine=u_f)
        near, far = NEAR, FAR

    def step():
        o, d, z, pts = ops.raygen_sample(c2w, FOCAL, FOCAL, W * 0.5, H * 0.5, H, W, pix, z_lin, u_c)
        loss = train_step(model, o, d, gt, near, far, N_COARSE, chunk_rays=chunk_rays, buckets=buckets, samples=(z, pts), **kw)
        opt.step()
        return loss

    step.opt = opt
    step.buckets = buckets
    if kind == "ngp" and occupancy:
        step.occupancy = occ_info
    return step, model


def dist_info(world, backend, dev, buckets):
    """Who took part in the run, for the reader of the JSON line: the process group's own world size and backend, the
    RCCL version, every rank's device (gathered, not assumed) and the bytes each rank's gradient all-reduces carry per
    step."""
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": int(os.environ.get("RANK", 0)), "device": dev.index, "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None}
    info = {"world": 1, "backend": None, "devices": [mine]}
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        info = {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": everyone}
    try:
        info["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info["rccl"] = None
    flats = getattr(buckets, "flats", [])
    info["allreduce_bytes_per_step"] = [4 * f.numel() for f in flats] if world > 1 else []
    info["allreduce_note"] = ("one sum all-reduce per gradient bucket (flat fp32 buffer) per step, issued under the last ray "
                              "chunk's backward; no other collective in the step" if world > 1 else "single rank: no collective")
    return info


def build_ngp_step(rank, rays, dev, chunk_rays, world, use_graph, occupancy=None):
    """The hash-grid step; with use_graph captured into a HIP graph and replayed where that applies: one GPU (no
    collective inside the step) and one ray chunk.  Returns (step, model, info for the result's config).
    occupancy: the step skips the samples in empty cells of a synthetic occupancy grid (a ball of occupied cells, sized
    so that about this fraction of the batch's samples is kept).  Replayed from a graph, the number of kept samples stays
    on the device and every tensor keeps its full size; issued launch by launch (--no-graph, several GPUs, ray chunks) the
    count is read back and the tensors shrink to it."""
    if occupancy:
        single = world == 1 and (chunk_rays is None or chunk_rays >= rays) and rays <= 65536
        if use_graph and single:
            # the number of kept samples stays in device memory (skip_empty="device"): fixed sizes, nothing read back
            try:
                from nerf_amd.graph import GraphedStep
                eager, model = build_step("ngp", rank, rays, dev, chunk_rays, capturable=True, occupancy=occupancy, skip_mode="device")
                graphed = GraphedStep(eager, warmup=3, optimizers=[eager.opt])
                graphed.buckets = eager.buckets
                return graphed, model, {"hip_graph": True, "skip_empty": "device", **eager.occupancy}
            except Exception as e:   # the same kernels run eagerly; the line says so
                print("hip graph capture of the skipping step failed, running eagerly: %r" % (e,), file=sys.stderr)
        step, model = build_step("ngp", rank, rays, dev, chunk_rays, occupancy=occupancy)
        return step, model, {"hip_graph": False, "skip_empty": True, **step.occupancy}
    if use_graph and world == 1 and (chunk_rays is None or chunk_rays >= rays) and rays <= 65536:
        try:
            from nerf_amd.graph import GraphedStep
            eager, model = build_step("ngp", rank, rays, dev, chunk_rays, capturable=True)
            graphed = GraphedStep(eager, warmup=3, optimizers=[eager.opt])
            graphed.buckets = eager.buckets
            return graphed, model, {"hip_graph": True}
        except Exception as e:   # the same kernels run eagerly; the line says so
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            step, model = build_step("ngp", rank, rays, dev, chunk_rays)
            return step, model, {"hip_graph": False, "hip_graph_error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    step, model = build_step("ngp", rank, rays, dev, chunk_rays)
    return step, model, {"hip_graph": False}


def timed_steps(step, steps, warmup, world, dev):
    import torch.distributed as dist

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, float(loss.detach())


def gpu_main(args):
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # NERF_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share cards)
    backend = os.environ.get("NERF_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # first collective from this thread: communicator and its stream exist before any gradient hook uses them
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    # one GPU: the 4096-ray batch the headline metric is quoted on; data parallel: 65,536 rays per GPU
    rays = args.rays if args.rays else (RAYS_SINGLE if world == 1 else RAYS_DP)
    chunk = args.chunk_rays
    graph_info = {}
    if args.model == "ngp":
        step, model, graph_info = build_ngp_step(rank, rays, dev, chunk, world, (not args.no_graph), args.occupancy)
    else:
        step, model = build_step(args.model, rank, rays, dev, chunk)
    dt, final_loss = timed_steps(step, args.steps, args.warmup, world, dev)
    who = dist_info(world, backend, dev, getattr(step, "buckets", None))   # a collective: every rank calls it
    secondary = secondary_skip = None
    if args.model == "vanilla" and not args.no_secondary:
        # the hash-grid configuration, same camera and batch size, as a second result in the same line
        del step
        torch.cuda.empty_cache()
        step2, ngp, graph2 = build_ngp_step(rank, rays, dev, chunk, world, (not args.no_graph))
        n2 = max(5, args.steps)
        dt2, loss2 = timed_steps(step2, n2, 3, world, dev)
        if rank == 0:
            secondary = {"metric": "rays/sec fwd+bwd (hash-grid model, 64 samples)", "value": world * rays * n2 / dt2,
                         "unit": "rays/s", "n_gpus": world, "steps": n2, "ms_per_step": 1e3 * dt2 / n2, "dtype": "f32",
                         "config": {"workload": NGP_WORKLOAD % rays, "rays_per_gpu": rays, "final_loss": loss2, **graph2},
                         "roofline": ngp_roofline(ngp, min(rays, 65536), dev)}
        # the same step skipping the samples in empty cells of a synthetic occupancy grid (a third result, beside the others)
        del step2, ngp
        torch.cuda.empty_cache()
        step3, _, info3 = build_ngp_step(rank, rays, dev, chunk, world, (not args.no_graph), occupancy=args.occupancy or 0.2)
        dt3, loss3 = timed_steps(step3, n2, 3, world, dev)
        if rank == 0:
            secondary_skip = {"metric": "rays/sec fwd+bwd (hash-grid model, 64 samples, empty samples skipped)",
                              "value": world * rays * n2 / dt3, "unit": "rays/s", "n_gpus": world, "steps": n2,
                              "ms_per_step": 1e3 * dt3 / n2, "dtype": "f32",
                              "config": {"workload": (NGP_WORKLOAD % rays) + "; samples in empty cells of a synthetic 32^3 occupancy "
                                                     "grid (a ball of occupied cells keeping %.2f of the samples) are dropped before the model: "
                                                     "same render as zeroing their densities" % (args.occupancy or 0.2),
                                         "rays_per_gpu": rays, "final_loss": loss3, **info3}}
        # ... and both at the data-parallel batch, where the step is bound by the GPU and not by its 25 launches
        if world == 1 and rays < 65536:
            del step3
            torch.cuda.empty_cache()
            big = 65536
            step4, _, info4 = build_ngp_step(rank, big, dev, None, world, (not args.no_graph))
            dt4, _ = timed_steps(step4, 5, 2, world, dev)
            del step4
            torch.cuda.empty_cache()
            step5, _, info5 = build_ngp_step(rank, big, dev, None, world, (not args.no_graph), occupancy=args.occupancy or 0.2)
            dt5, _ = timed_steps(step5, 5, 2, world, dev)
            del step5
            torch.cuda.empty_cache()
            secondary_skip["at_65536_rays"] = {"ms_per_step_all_samples": 1e3 * dt4 / 5, "rays_per_s_all_samples": big * 5 / dt4,
                                               "ms_per_step_empty_skipped": 1e3 * dt5 / 5, "rays_per_s_empty_skipped": big * 5 / dt5,
                                               "samples_kept": info5.get("samples_kept"), "steps": 5, "hip_graph": info5.get("hip_graph"),
                                               "note": "the same two steps at 65,536 rays per GPU (the data-parallel batch)"}

    result = None
    if rank == 0 and args.model == "ngp":
        result = {
            "metric": "rays/sec fwd+bwd (hash-grid model, 64 samples)", "value": world * rays * args.steps / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (NGP_WORKLOAD % rays) + ("; samples in empty cells of a synthetic occupancy grid (a ball of "
                                                                "occupied cells keeping %.2f of the samples) skipped" % args.occupancy if args.occupancy else ""),
                       "rays_per_gpu": rays, "parallelism": "ray-sharded dp%d" % world,
                       "final_loss": final_loss, **graph_info},
            "roofline": ngp_roofline(model, min(rays, 65536), dev),
            "dist": who,
        }
    elif rank == 0:
        roof = kernel_roofline(model, min(rays, RAYS_SINGLE) if chunk is None else min(rays, chunk), dev, probe=not args.no_probe)
        result = {
            "metric": "rays/sec fwd+bwd (lego 800x800, 64 coarse + 128 fine)",
            "value": world * rays * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (operands split into 2 f16 pieces each, 3 f16 MFMAs per product, fp32 accumulate: fp32-accurate)",
            "data": "synthetic",
            "config": {"workload": "vanilla NeRF 8x256 coarse+fine, Blender-lego-shaped 800x800 camera, "
                                   "64 coarse + 128 fine samples, %d-ray batch per GPU%s, one training step "
                                   "(fused ray-gen + sampling, MLPs, compositing, backward, %sAdam)"
                                   % (rays, "" if rays <= RAYS_SINGLE else " in ray chunks with gradient accumulation",
                                      "" if world == 1 else "bucketed RCCL all-reduce of the gradients, "),
                       "rays_per_gpu": rays, "parallelism": "ray-sharded dp%d" % world,
                       "final_loss": final_loss},
            "roofline": roof,
            "dist": who,
        }
        if secondary is not None:
            result["secondary"] = secondary
        if secondary_skip is not None:
            result["secondary_skip_empty"] = secondary_skip
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(args.cpu_rays)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


MLP_SOURCES = ("common.h", "mlp_common.h", "mlp_shared.hip", "mlp_vanilla.hip")


def mlp_source_sha():
    """sha256 over the sources of the three vanilla-MLP kernels: what a set of hardware-counter passes belongs to."""
    import hashlib
    h = hashlib.sha256()
    for name in MLP_SOURCES:
        with open(os.path.join(ROOT, "nerf_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _time_kernel(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def sustained_mfma_probe():
    """What the card's matrix cores sustain on real data, measured now: tools/microbench/mfma_sustained (a loop of
    nothing but 32x32x16 f16 MFMAs on random register operands, every CU, ~80 ms per launch) run as a child process.
    The nominal 2.5 PFLOP/s is 2.4 GHz; under its power limit the card holds about 1.66 GHz in that loop.  None when
    the probe has not been built."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "microbench", "mfma_sustained")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=60).stdout
        rows = [ln.split() for ln in out.splitlines() if ln.startswith("random")]
        tflops = [float(r[r.index("TFLOP/s") - 1]) for r in rows]
        mhz = [float(r[r.index("MHz") - 1]) for r in rows]
        if not tflops:
            return None
        return {"f16_tflops": tflops[-1], "shader_clock_mhz": mhz[-1], "frac_of_nominal": tflops[-1] / PEAK_F16_MATRIX_TFLOPS,
                "launches_tflops": tflops,
                "what": "v_mfma_f32_32x32x16_f16 back to back on random normal register operands, one wave per SIMD on every "
                        "CU, no memory traffic, last of 4 launches of ~80 ms (tools/microbench/mfma_sustained.hip); with "
                        "all-zero operands the same loop runs at 2.39 GHz and 0.99 of nominal"}
    except (OSError, ValueError, subprocess.SubprocessError):
        return None


def kernel_roofline(model, rays, dev, probe=True):
    """Time the three MLP kernels alone (HIP events on the launch stream) at the
    fine-pass size and report the one that takes most of the step."""
    from nerf_amd import ops
    P = rays * (N_COARSE + N_FINE)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(P, 3, generator=g) * 4 - 2).to(dev)
    d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1).to(dev)
    d_rgb = torch.randn(P, 3, generator=g).to(dev)
    d_sig = torch.randn(P, 1, generator=g).to(dev)
    params = model._params()
    packed = ops.vanilla_pack(params, False)
    rgb, sigma, ws = ops.vanilla_fwd_raw(x, d, N_COARSE + N_FINE, packed, True)
    ws2 = ops.vanilla_ws2(P, dev)
    grads = ops.zero_grads_like([p.shape for p in params], dev)
    lib_ws = ws  # keep alive

    def fwd():
        from nerf_amd import _lib as L
        L.check(L.lib().nerf_mlp_vanilla_fwd(L.ptr(x), L.ptr(d), N_COARSE + N_FINE, P, L.ptr(packed), L.ptr(rgb),
                                             L.ptr(sigma), L.ptr(lib_ws), L.stream_of(x)), "fwd")

    t_fwd = _time_kernel(fwd)
    t_bd = _time_kernel(lambda: ops.vanilla_bwd_raw(d_rgb, d_sig, packed, rgb, sigma, ws, ws2, False, grads, phases=1))
    t_wg = _time_kernel(lambda: ops.vanilla_bwd_raw(d_rgb, d_sig, packed, rgb, sigma, ws, ws2, False, grads, phases=2))
    kernels = {
        "vanilla_fwd_kernel": (t_fwd, FLOP_FWD, BYTES_FWD),
        "vanilla_bwd_data_kernel": (t_bd, FLOP_BWD_DATA, BYTES_BWD_DATA),
        "wgrad_kernel": (t_wg, FLOP_WGRAD, BYTES_WGRAD),
    }
    name = max(kernels, key=lambda k: kernels[k][0])
    t, flop, nbytes = kernels[name]
    tflops = flop * P / t / 1e12
    gbs = nbytes * P / t / 1e9
    peak_mfma = PEAK_F16_MATRIX_TFLOPS / MFMA_PER_PRODUCT
    # HBM bytes per launch of that kernel from the committed PMC passes (measured
    # offline with rocprofv3 --pmc: a counter run cannot be nested inside this one)
    # (they belong to one state of the kernel sources: passes collected from other sources are not reported)
    traffic, busy, stale = None, {}, None
    try:
        with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as fh:
            tr = json.load(fh)
        stale = tr.get("source_sha256") != mlp_source_sha()
        if not stale and tr.get("points_per_launch") == P and name in tr:
            traffic = tr[name]["hbm_bytes"]
            busy = {k: tr[k]["mfma_busy"] for k in kernels if k in tr and "mfma_busy" in tr[k]}
    except (OSError, ValueError, KeyError):
        traffic = None
    # the kernel is bounded by whichever resource it uses the larger fraction of
    hbm_bound = gbs / PEAK_HBM_GBS >= tflops / peak_mfma
    per_kernel = {k: {"ms": v[0] * 1e3, "tflops": v[1] * P / v[0] / 1e12, "mfma_frac": v[1] * P / v[0] / 1e12 / peak_mfma,
                      "gbs": v[2] * P / v[0] / 1e9, "hbm_frac": v[2] * P / v[0] / 1e9 / PEAK_HBM_GBS} for k, v in kernels.items()}
    sustained = sustained_mfma_probe() if probe else None
    if sustained:
        for v in per_kernel.values():
            v["frac_of_sustained_mfma"] = v["tflops"] * MFMA_PER_PRODUCT / sustained["f16_tflops"]
    return {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
            "achieved": gbs if hbm_bound else tflops, "peak": PEAK_HBM_GBS if hbm_bound else peak_mfma,
            "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": gbs / PEAK_HBM_GBS if hbm_bound else tflops / peak_mfma, "traffic": traffic, "points_per_launch": P,
            "note": "the slowest of the three MLP kernels, against the resource it uses most: HBM (achieved = algorithmic bytes "
                    "per point x points / HIP-event time, peak 8 TB/s) or the matrix cores (achieved = algorithmic fp32-equivalent "
                    "FLOP/s, peak = 2500 TFLOP/s dense f16 / 3 f16 MFMAs per fp32 product; the exact-f32 MFMA peak is 157.3)",
            "mfma_frac": tflops / peak_mfma, "hbm_frac": gbs / PEAK_HBM_GBS, "executed_f16_tflops": tflops * MFMA_PER_PRODUCT,
            "fp32_equivalent_tflops": tflops,
            "frac_of_six_mfma_roofline": tflops / (PEAK_F16_MATRIX_TFLOPS / 6),
            "note_six": "an earlier version formed a product from six bf16 MFMAs (roofline 416.7 fp32-equivalent TFLOP/s, kernels at "
                        "0.49-0.54 of it); halving the MFMAs per product halves the executed-MFMA fraction of the same time, so "
                        "the fp32-equivalent rate against that earlier roofline is given for comparison",
            "mfma_busy_pmc": busy or None, "traffic_stale": stale,
            "note_busy": "SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD of the committed rocprofv3 counter pass "
                         "(profiles/roofline_traffic.json): the share of the kernel's own clocks (1.6-2.0 GHz under this load, not "
                         "the 2.4 GHz of the peak) during which the matrix pipe was busy",
            "sustained_mfma_measured": sustained,
            "frac_of_sustained_mfma": tflops * MFMA_PER_PRODUCT / sustained["f16_tflops"] if sustained else None,
            "note_sustained": "frac and peak above are against the nominal 2.5 PFLOP/s (2.4 GHz); sustained_mfma_measured is what a "
                              "loop of nothing but f16 MFMAs on random operands reaches on this card in this run (the power limit "
                              "lowers the clock), and frac_of_sustained_mfma the kernel's executed f16 MFMA rate against that",
            "all_kernels": per_kernel,
            "all_kernels_ms": {k: v[0] * 1e3 for k, v in kernels.items()}}


def ngp_roofline(model, rays, dev):
    """The table-gradient scatter is what the hash-grid step spends most of its time in: algorithmic
    bytes (1,024 B of float atomics per point) over its HIP-event time, against the HBM peak."""
    from nerf_amd import _lib as L
    P = rays * N_COARSE
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_enc = torch.randn(P, 32, generator=g).to(dev)
    tabs = [e.weight.detach() for e in model.hash_encoder.embeddings]
    grads = [torch.zeros_like(t) for t in tabs]
    res = L.int_array(model.hash_encoder.resolutions)
    sizes = L.int_array([int(t.shape[0]) for t in tabs])
    ptrs = L.ptr_array(grads)
    from nerf_amd import ops
    work, nbytes = ops.scatter_workspace(P, [int(t.shape[0]) for t in tabs], dev)
    t = _time_kernel(lambda: L.check(L.lib().nerf_hashgrid_bwd(L.ptr(x), P, L.ptr(d_enc), res, sizes, 16, ptrs, L.ptr(work),
                                                               nbytes, L.stream_of(x)), "hashgrid_bwd"))
    achieved = 1024.0 * P / t / 1e9
    return {"bound": "hbm", "kernel": "hashgrid table-gradient scatter (bin + reduce + small-level kernels)",
            "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
            "points_per_launch": P, "ms": t * 1e3,
            "note": "achieved = 1,024 B of table updates per point (16 levels x 8 corners x 8 B) / time of the whole scatter"}


def _cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rays, rays_single=128):
    """The reference's op sequence (CPU PyTorch restatement in oracle/) on the host
    cores: same camera, same sampling, two 8x256 MLPs evaluated in 2048-point
    chunks, compositing, loss, backward, Adam -- on a bounded ray sample, with all
    threads and with one."""
    from oracle import nerf_oracle as O
    threads = torch.get_num_threads()

    def run(n_rays):
        torch.manual_seed(0)
        coarse = O.NeRFOracle()
        torch.manual_seed(1)
        fine = O.NeRFOracle()
        opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
        g = torch.Generator().manual_seed(1)
        pix = torch.randperm(H * W, generator=g)[:n_rays]
        g2 = torch.Generator().manual_seed(2)
        u_c = torch.rand(n_rays, N_COARSE, generator=g2)
        u_f = torch.rand(n_rays, N_FINE, generator=g2)
        gt = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(3))
        c2w = lego_pose(0)
        o_all, d_all = O.get_rays(c2w, (FOCAL, FOCAL), H, W)   # the reference generates the whole image's rays

        def step():
            o, d = o_all.reshape(-1, 3)[pix], d_all.reshape(-1, 3)[pix]
            c_map, w, z, fine_out = O.render_rays(coarse, o, d, NEAR, FAR, N_COARSE, t_rand=u_c, chunk=2048,
                                                  white_bkgd=True, fine_model=fine, u_fine=u_f)
            loss = torch.nn.functional.mse_loss(c_map, gt) + torch.nn.functional.mse_loss(fine_out[0], gt)
            opt.zero_grad()
            loss.backward()
            opt.step()

        step()
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        return n_rays / sorted(times)[1]

    all_threads = run(rays)
    torch.set_num_threads(1)
    try:
        one_thread = run(rays_single)
    finally:
        torch.set_num_threads(threads)
    return {"value": all_threads, "unit": "rays/s", "cores": threads, "kind": "port",
            "single_thread": {"value": one_thread, "unit": "rays/s", "cores": 1, "sample": "%d-ray batch" % rays_single},
            "cpu": _cpu_model_name(), "host_cpus": os.cpu_count(), "torch": torch.__version__, "rays": rays,
            "sample": "%d-ray batch of the same workload (64c+128f, two 8x256 MLPs in 2048-point chunks, fwd+bwd+Adam), "
                      "median of 3 steps after 1 warm-up" % rays}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=0, help="rays per GPU per step (default: 4096 on one GPU, 65536 per GPU on several)")
    ap.add_argument("--chunk-rays", type=int, default=None, help="rays per forward/backward chunk (default: what fits 32 GiB of workspace)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the hash-grid result that rides along as 'secondary'")
    ap.add_argument("--model", choices=["vanilla", "ngp"], default="vanilla",
                    help="vanilla = the headline workload; ngp = the hash-grid variant (secondary line)")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="ray batch of the CPU baseline's all-thread leg (the batch the metric is quoted on)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the sustained-MFMA probe (a child process: under a profiler it would be traced as well)")
    ap.add_argument("--occupancy", type=float, default=None,
                    help="hash-grid model: skip the samples in empty cells of a synthetic occupancy grid (a ball of occupied cells) "
                         "that keeps about this fraction of the samples; rides along as 'secondary_skip_empty' at 0.2 by default")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue the hash-grid step launch by launch instead of replaying it from a HIP graph (one GPU, one ray "
                         "chunk): 0.83 against 0.74 ms per 4096-ray step -- at 23 launches per step the host is part of it")
    args = ap.parse_args()
    if os.environ.get("NERF_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack and exit after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NERF_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run (nothing has touched the GPU yet)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    gpu_main(args)


if __name__ == "__main__":
    main()
