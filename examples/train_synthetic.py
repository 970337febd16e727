#!/usr/bin/env python3
"""The reference's synthetic-scene training loop (src/train_nerf.py:49-204) on this package: 40 cameras on a ring
around an analytic radiance field, one 100x100 view per step, Adam 5e-4, gradient clipping at 1, ExponentialLR 0.98
per epoch, and the per-epoch log of loss / SSIM / PSNR.  Everything stays on the GPU; the log is the only thing that
comes back.

    python examples/train_synthetic.py --epochs 21 --out /tmp/run     # ~1 s per epoch of 40 views on an MI355X
"""
import argparse
import csv
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_amd import ops  # noqa: E402
from nerf_amd.Camera import Camera  # noqa: E402
from nerf_amd.models.NeRF import NeRF  # noqa: E402
from nerf_amd.graph import GraphedStep  # noqa: E402
from nerf_amd.optim import FusedAdam  # noqa: E402
from nerf_amd.render import nerf_loss, psnr, render_rays, ssim  # noqa: E402
from nerf_amd.utils.get_rays import evaluate_fake_radiance_field_complex  # noqa: E402


DENSITY_BOOST = 5.0


def ring_views(dev, H, W, focal, near, far, n_samples, n_views=40):
    """Ground-truth images of the analytic field from the reference's camera ring (src/train_nerf.py:49-68).
    ``Camera`` shoots its rays along -forward (src/Camera.py:61-68), i.e. away from the look-at target: a camera that is
    to see the origin aims at the mirrored point.  The field's soft densities (at most 1) are boosted for contrast."""
    views = []
    up = torch.tensor([0.0, 0.0, 1.0], device=dev)
    for a in torch.linspace(0, 2 * math.pi, n_views).tolist():
        eye = torch.tensor([2 * math.cos(a), 2 * math.sin(a), 2.0], device=dev)
        cam = Camera(eye=eye, target=2 * eye, focal=(focal, focal), H=H, W=W, up=up)
        rays_o, rays_d = cam.get_rays()
        pts, z = Camera.sample_points_along_rays(rays_o, rays_d, near, far, n_samples, perturb=False)
        rgb, sigma = evaluate_fake_radiance_field_complex(pts)
        image, _ = Camera.volume_render(rgb, sigma * DENSITY_BOOST, z)
        views.append((rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), image.detach()))
    return views


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=21)
    ap.add_argument("--size", type=int, default=100, help="image side and focal length (the reference: 100)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--out", default=None, help="directory for log.csv and the final weights (default: no files)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--importance", type=int, default=0,
                    help="fine samples per ray drawn from the coarse pass's weights and a second NeRF evaluated at all of them "
                         "(the reference's synthetic loop has a coarse pass only: 0)")
    ap.add_argument("--graph", action="store_true", help="capture the step into a HIP graph and replay it")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    H = W = args.size
    near, far = 1.0, 5.0
    model = NeRF().to(dev)
    fine = NeRF().to(dev) if args.importance else None
    params = list(model.parameters()) + (list(fine.parameters()) if fine else [])
    opt = FusedAdam(params, lr=5e-4, capturable=args.graph)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.98)
    views = ring_views(dev, H, W, float(args.size), near, far, args.samples)
    z_lin = torch.linspace(near, far, args.samples).to(dev)
    # the step reads its view from these tensors (a captured step must; an eager one may)
    s_o, s_d, s_img = torch.empty(H * W, 3, device=dev), torch.empty(H * W, 3, device=dev), torch.empty(H * W, 3, device=dev)

    def one_step():
        opt.zero_grad(set_to_none=False)
        out = render_rays(model, s_o, s_d, near, far, args.samples, z_lin=z_lin, fine_model=fine, n_importance=args.importance)
        loss = nerf_loss(out, s_img)       # MSE of the coarse colour, plus the fine colour's when there is a fine pass
        loss.backward()
        ops.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss, out["rgb_fine" if fine else "rgb"].detach()

    def load(i):
        rays_o, rays_d, image = views[i % len(views)]
        s_o.copy_(rays_o), s_d.copy_(rays_d), s_img.copy_(image.reshape(-1, 3))

    step = GraphedStep(one_step, warmup=3, optimizers=[opt], prepare=load) if args.graph else one_step
    rows = []
    torch.cuda.synchronize()
    t0 = time.time()
    for epoch in range(args.epochs):
        loss_sum = torch.zeros((), device=dev)
        ssim_sum = torch.zeros((), device=dev, dtype=torch.float64)
        psnr_sum = torch.zeros((), device=dev)
        for i, (rays_o, rays_d, image) in enumerate(views):
            load(i)
            loss, rgb = step()
            with torch.no_grad():
                pred = rgb.view(H, W, 3).clamp(0, 1)
                loss_sum += loss.detach()
                ssim_sum += ssim(image.clamp(0, 1), pred)
                psnr_sum += psnr(image.clamp(0, 1), pred)
        sched.step()
        n = len(views)
        row = (epoch, float(loss_sum) / n, float(ssim_sum) / n, float(psnr_sum) / n)   # one read-back per epoch
        rows.append(row)
        print("[epoch %d/%d] loss %.5f | ssim %.4f | psnr %.3f dB | lr %.2e | %.1f s" %
              (epoch + 1, args.epochs, row[1], row[2], row[3], sched.get_last_lr()[0], time.time() - t0), flush=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = args.epochs * len(views)
    print("%d steps of %d rays x %d samples in %.2f s: %.2f ms per step, %.0f rays/s" %
          (steps, H * W, args.samples, dt, 1e3 * dt / steps, steps * H * W / dt))
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        with open(os.path.join(args.out, "log.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["epoch", "loss", "ssim", "psnr"])
            w.writerows(rows)
        torch.save(model.state_dict(), os.path.join(args.out, "nerf_final.pth"))   # loads into the reference's NeRF as is


if __name__ == "__main__":
    main()
