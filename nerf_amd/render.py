"""Ray-batch rendering and the training step built from the HIP ops.

This is the counterpart of the reference's ``render_image`` helpers
(src/train_nerf_llff.py:31-64, src/train_nerf_skow.py:66-100,
src/train_nerf_ngp.py:80-126): sample -> evaluate the model on every sample ->
composite.  Their Python chunk loop is gone (the fused MLP kernel tiles the
points itself) and an optional second, importance-sampled pass is added
(hierarchical sampling is not in the reference).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def _eval_model(model, pts, view_dirs, n_per_ray):
    """pts [R,N,3], view_dirs [R,3] -> rgb [R,N,3], sigma [R,N]."""
    R, N = pts.shape[0], pts.shape[1]
    if hasattr(model, "_params") and hasattr(model, "_skip_h_first"):
        # fused vanilla MLP: one direction row per ray serves all of its samples
        rgb, sigma = ops.vanilla_mlp(pts.reshape(-1, 3), view_dirs, model._params(), model._skip_h_first, dirs_div=N)
    elif hasattr(model, "_tables"):
        # fused hash-grid model: same one-row-per-ray trick
        rgb, sigma = model(pts.reshape(-1, 3), view_dirs, dirs_div=N)
    else:
        dirs = view_dirs[:, None, :].expand(R, N, 3).reshape(-1, 3)
        rgb, sigma = model(pts.reshape(-1, 3), dirs)
    return rgb.view(R, N, 3), sigma.reshape(R, N)


def render_rays(model, rays_o, rays_d, near: float, far: float, n_samples: int, perturb: bool = True,
                t_rand: Optional[torch.Tensor] = None, white_bkgd: bool = False, fine_model=None,
                n_importance: int = 0, u_fine: Optional[torch.Tensor] = None):
    """Render a batch of rays [R,3].

    Returns a dict with ``rgb`` [R,3], ``weights`` [R,N], ``z_vals`` [R,N],
    ``depth`` [R] of the coarse pass and, when ``fine_model`` is given,
    ``rgb_fine``, ``weights_fine``, ``z_fine`` (the merged, sorted depths) and
    ``depth_fine``.  ``t_rand`` [R,N] / ``u_fine`` [R,n_importance] inject the
    uniform draws (drawn on the GPU when omitted).
    """
    R = rays_o.shape[0]
    dev = rays_o.device
    z_lin = torch.linspace(near, far, n_samples, device=dev)
    if perturb and t_rand is None:
        t_rand = torch.rand(R, n_samples, device=dev)
    z, pts = ops.stratified_sample(rays_o, rays_d, z_lin, t_rand if perturb else None)
    view_dirs = rays_d / rays_d.norm(dim=-1, keepdim=True)   # as render_image does (src/train_nerf_llff.py:42)
    rgb, sigma = _eval_model(model, pts, view_dirs, n_samples)
    c_map, w, depth = ops.composite(rgb, sigma, z, white_bkgd)
    out = {"rgb": c_map, "weights": w, "z_vals": z, "depth": depth}
    if fine_model is not None and n_importance > 0:
        if u_fine is None:
            u_fine = torch.rand(R, n_importance, device=dev)
        _, _, z_all, pts_f = ops.pdf_sample(rays_o, rays_d, z, w, u_fine)
        rgb_f, sigma_f = _eval_model(fine_model, pts_f, view_dirs, n_samples + n_importance)
        f_map, w_f, depth_f = ops.composite(rgb_f, sigma_f, z_all, white_bkgd)
        out.update({"rgb_fine": f_map, "weights_fine": w_f, "z_fine": z_all, "depth_fine": depth_f})
    return out


def psnr(x, y):
    """-10 log10(mse), the reference's definition (src/train_nerf_llff.py:18-19)."""
    return -10.0 * torch.log10(torch.mean((x - y) ** 2))


def nerf_loss(out: dict, target: torch.Tensor) -> torch.Tensor:
    """MSE of the coarse colour (src/train_nerf_llff.py:116) plus, when a fine
    pass was rendered, the MSE of the fine colour."""
    loss = torch.nn.functional.mse_loss(out["rgb"], target)
    if "rgb_fine" in out:
        loss = loss + torch.nn.functional.mse_loss(out["rgb_fine"], target)
    return loss
