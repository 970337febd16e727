#!/usr/bin/env python3
"""The hash-grid training loop of the reference (src/train_nerf_ngp.py:41-62, 80-126, 190-260) on this package, fed by
the analytic radiance field shrunk into the unit cube instead of a captured scene (none ships with either repo):
ray batches across all views from a GPU-resident buffer, Adam(1e-2, eps 1e-15), MSE + 0.05 * mean |1 - sum w|,
ReduceLROnPlateau handing over to CosineAnnealingLR, an occupancy grid refreshed from the model's own densities, and --
once the grid exists -- the samples in its empty cells skipped.

    python examples/train_ngp_synthetic.py --steps 1500
    python examples/train_ngp_synthetic.py --steps 1500 --graph    # the step replayed from a HIP graph: the number of
                                                                   # samples that survive the grid stays on the device
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd.Camera import Camera  # noqa: E402
from nerf_amd.data.ray_buffer import RayBuffer  # noqa: E402
from nerf_amd.dist import GradBuckets  # noqa: E402
from nerf_amd.models.FastNGPNeRF import FastNGPNeRF  # noqa: E402
from nerf_amd.optim import FusedAdam  # noqa: E402
from nerf_amd.graph import GraphedStep  # noqa: E402
from nerf_amd.render import get_density_grid, ngp_loss, psnr, render_image, train_step  # noqa: E402
from nerf_amd.utils.get_rays import evaluate_fake_radiance_field_complex  # noqa: E402

SHRINK = 2.5            # the field's torus (radius 1.3 overall) fits the unit cube at 1 / 2.5
NEAR, FAR = 0.3, 1.6     # the object starts 0.42 from the cameras: space only one camera sees is left out (a model paints
                         # each training image onto density right in front of its camera otherwise)
BOOST = 5.0 * SHRINK    # density per unit length grows as lengths shrink


def views_on_ring(dev, n_views, H, W, focal, n_samples, height=0.5, radius=0.8):
    """(poses [n,3,4], intrinsics [n,4], images [n,H,W,3]).  ``Camera`` shoots along -forward (src/Camera.py:61-68):
    a camera that is to see the origin aims at the mirrored point."""
    up = torch.tensor([0.0, 0.0, 1.0], device=dev)
    poses, images = [], []
    for k in range(n_views):
        a = 2 * math.pi * k / n_views
        eye = torch.tensor([radius * math.cos(a), radius * math.sin(a), height], device=dev)
        cam = Camera(eye=eye, target=2 * eye, focal=(focal, focal), H=H, W=W, up=up)
        rays_o, rays_d = cam.get_rays()
        pts, z = Camera.sample_points_along_rays(rays_o, rays_d, NEAR, FAR, n_samples, perturb=False)
        rgb, sigma = evaluate_fake_radiance_field_complex(pts * SHRINK)
        image, _ = Camera.volume_render(rgb, sigma * BOOST, z)
        poses.append(cam.c2w)
        images.append(image)
    intr = torch.tensor([focal, focal, W * 0.5, H * 0.5], device=dev).expand(n_views, 4)
    return torch.stack(poses), intr, torch.stack(images)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--size", type=int, default=100)
    ap.add_argument("--views", type=int, default=41, help="the last one is held out")
    ap.add_argument("--grid-every", type=int, default=250, help="steps between refreshes of the occupancy grid")
    ap.add_argument("--grid-res", type=int, default=32)
    ap.add_argument("--reg", type=float, default=0.0,
                    help="weight of the opacity term mean |1 - sum w| (the reference: 0.05, for captured scenes that are opaque along "
                         "every ray; this field is mostly empty space, where the term fills the background with black density)")
    ap.add_argument("--graph", action="store_true", help="capture the step into a HIP graph and replay it")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H = W = args.size
    n_samples = 64
    poses, intr, images = views_on_ring(dev, args.views, H, W, float(args.size), 128)
    buf = RayBuffer(images[:-1], poses[:-1], intr[:-1], dev)
    held_cam = Camera(eye=None, target=None, focal=(float(args.size),) * 2, H=H, W=W, c2w=poses[-1])
    held_o, held_d = held_cam.get_rays()

    model = FastNGPNeRF().to(dev)
    # gradients land in two flat buffers (tables, everything else) instead of one tensor per parameter: two fills per
    # step instead of 25, no per-parameter accumulation kernels, and on several GPUs the buffers are what is all-reduced
    buckets = GradBuckets([list(model.hash_encoder.parameters()),
                           [p for n, p in model.named_parameters() if not n.startswith("hash_encoder")]])
    opt = FusedAdam(buckets.parameters(), lr=1e-2, eps=1e-15, capturable=args.graph)
    loss_fn = lambda out, t: ngp_loss(out, t, reg=args.reg)   # noqa: E731
    plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5, patience=2)
    cosine = None
    grid = None
    window, window_n = torch.zeros((), device=dev), 0
    graphed = None
    if args.graph:
        # the captured step reads these tensors; every iteration copies its batch, its uniforms and the current grid into them
        s_o, s_d, s_t = (torch.empty(args.rays, 3, device=dev) for _ in range(3))
        s_u = torch.empty(args.rays, n_samples, device=dev)
        s_grid = torch.ones(args.grid_res ** 3, dtype=torch.uint8, device=dev)   # everything occupied until the first refresh
        z_lin = torch.linspace(NEAR, FAR, n_samples).to(dev)

        def load(_=None):
            o, d, t = buf.next_batch(args.rays)
            s_o.copy_(o), s_d.copy_(d), s_t.copy_(t), s_u.uniform_()

        def captured():
            loss = train_step(model, s_o, s_d, s_t, NEAR, FAR, n_samples, loss_fn=loss_fn, buckets=buckets, t_rand=s_u,
                              z_lin=z_lin, density_grid=s_grid, grid_res=args.grid_res, skip_empty="device")
            opt.step()
            return loss

        graphed = GraphedStep(captured, warmup=3, optimizers=[opt], prepare=load)
    torch.cuda.synchronize()
    t0 = time.time()
    for step in range(1 + (graphed.steps_run if graphed else 0), args.steps + 1):
        if graphed:
            load()
            loss = graphed()
        else:
            rays_o, rays_d, target = buf.next_batch(args.rays)
            loss = train_step(model, rays_o, rays_d, target, NEAR, FAR, n_samples, loss_fn=loss_fn, buckets=buckets,
                              density_grid=grid, grid_res=args.grid_res, skip_empty=grid is not None)
            opt.step()
        window += loss.detach()
        window_n += 1
        if step % args.grid_every == 0 or step == args.steps:
            mean_loss = float(window) / window_n            # the only read-back of the loop
            window.zero_()
            window_n = 0
            if cosine is None:
                plateau.step(mean_loss)
                if step >= args.steps // 2:                 # the reference switches schedulers once (train_nerf_ngp.py:57-62)
                    cosine = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(1, (args.steps - step) // args.grid_every),
                                                                        eta_min=1e-4)
            else:
                cosine.step()
            grid = get_density_grid(model, dev, args.grid_res)
            if graphed:
                s_grid.copy_(grid)
            img = render_image(model, held_o, held_d, NEAR, FAR, n_samples=n_samples, grid_res=args.grid_res,
                               density_grid=grid, skip_empty=True)
            print("[step %d] loss %.5f | held-out psnr %.2f dB | grid %.1f %% occupied | lr %.2e | %.2f s" %
                  (step, mean_loss, float(psnr(images[-1].clamp(0, 1), img.clamp(0, 1))), 100.0 * float(grid.float().mean()),
                   opt.param_groups[0]["lr"], time.time() - t0), flush=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("%d steps of %d rays x %d samples in %.2f s: %.2f ms per step, %.2f M rays/s" %
          (args.steps, args.rays, n_samples, dt, 1e3 * dt / args.steps, args.steps * args.rays / dt * 1e-6))


if __name__ == "__main__":
    main()
