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


def xyz_to_idx(x, y, z, res: int = 32):
    """Flat cell index of the occupancy grid (src/utils/analyze_capture_results.py:230-231)."""
    return x + y * res + z * res * res


def get_density_grid(model, device, grid_res: int):
    """uint8 [grid_res^3] occupancy grid: 1 where ``model.query_density`` exceeds 1 at the cell's
    lattice point in [-1, 1]^3 (src/utils/analyze_capture_results.py:233-248)."""
    coords = torch.stack(torch.meshgrid(torch.arange(grid_res, device=device), torch.arange(grid_res, device=device),
                                        torch.arange(grid_res, device=device), indexing="ij"), dim=-1).view(-1, 3)
    grid = torch.zeros(grid_res ** 3, dtype=torch.uint8, device=device)
    with torch.no_grad():
        for i in range(0, coords.shape[0], 1 << 20):
            c = coords[i:i + (1 << 20)]
            pts = (c.float() / (grid_res - 1)) * 2 - 1
            mask = model.query_density(pts) > 1.0
            # the reference indexes with xyz_to_idx's default res=32 whatever grid_res is; kept
            grid[xyz_to_idx(c[:, 0], c[:, 1], c[:, 2])] |= mask.to(torch.uint8)
    return grid


def render_image(model, rays_o, rays_d, near: float, far: float, n_samples: int = 64, grid_res: Optional[int] = None,
                 density_grid: Optional[torch.Tensor] = None, white_bkgd: bool = False, fine_model=None,
                 n_importance: int = 0, rays_per_tile: int = 65536, return_depth: bool = False):
    """Full-image inference: rays [H,W,3] -> colour map [H,W,3] (and depth [H,W]).

    The counterpart of ``render_from_model`` (src/utils/analyze_capture_results.py:250-296): unperturbed
    samples, no autograd, sigma zeroed outside the occupancy grid when one is given.  The reference
    materialises all H*W*N points and loops over 16384-point chunks of them; here the loop is over tiles
    of rays and each tile runs sample -> model -> composite end to end.  With ``fine_model`` a second,
    importance-sampled pass follows (deterministic: evenly spaced CDF positions).
    """
    H, W = rays_o.shape[0], rays_o.shape[1]
    o_all, d_all = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    R = o_all.shape[0]
    dev = o_all.device
    out = torch.empty(R, 3, device=dev)
    depth = torch.empty(R, device=dev) if return_depth else None
    z_lin = torch.linspace(near, far, n_samples, device=dev)
    with torch.no_grad():
        for r0 in range(0, R, rays_per_tile):
            o, d = o_all[r0:r0 + rays_per_tile], d_all[r0:r0 + rays_per_tile]
            z, pts = ops.stratified_sample(o, d, z_lin, None)
            view_dirs = d / d.norm(dim=-1, keepdim=True)
            rgb, sigma = _eval_model(model, pts, view_dirs, n_samples)
            if density_grid is not None:
                sigma = _mask_sigma(sigma, pts, grid_res, density_grid)
            c_map, w, dep = ops.composite(rgb, sigma, z, white_bkgd)
            if fine_model is not None and n_importance > 0:
                u = torch.linspace(0.0, 1.0, n_importance, device=dev).expand(o.shape[0], n_importance).contiguous()
                _, _, z, pts = ops.pdf_sample(o, d, z, w, u)
                rgb, sigma = _eval_model(fine_model, pts, view_dirs, n_samples + n_importance)
                if density_grid is not None:
                    sigma = _mask_sigma(sigma, pts, grid_res, density_grid)
                c_map, w, dep = ops.composite(rgb, sigma, z, white_bkgd)
            out[r0:r0 + rays_per_tile] = c_map
            if return_depth:
                depth[r0:r0 + rays_per_tile] = dep
    img = out.view(H, W, 3)
    return (img, depth.view(H, W)) if return_depth else img


def _mask_sigma(sigma, pts, grid_res, density_grid):
    """sigma with the samples in empty occupancy cells zeroed (src/utils/analyze_capture_results.py:284-288)."""
    cell = 1.0 / grid_res
    x, y, z = ((pts.reshape(-1, 3) + 1) / 2 / cell).long().T.unbind(0)
    # samples outside [-1, 1)^3 index past the grid in the reference (an IndexError there); they are
    # clamped to the nearest stored cell here, since a device-side index fault would end the process
    idx = xyz_to_idx(x, y, z).clamp_(0, density_grid.numel() - 1)
    keep = density_grid[idx] == 1
    return torch.where(keep.view(sigma.shape), sigma, torch.zeros_like(sigma))


def render_image_sharded(model, rays_o, rays_d, near: float, far: float, group=None, **kw):
    """``render_image`` with the image rows split over the ranks of ``group`` (one process per GPU)
    and gathered at the end; every rank returns the whole image.  No other exchange is needed."""
    import torch.distributed as dist

    from .dist import gather_rows, shard_range
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return render_image(model, rays_o, rays_d, near, far, **kw)
    rank = dist.get_rank(group)
    H = rays_o.shape[0]
    lo, hi = shard_range(H, rank, world)
    part = render_image(model, rays_o[lo:hi], rays_d[lo:hi], near, far, **kw)
    sizes = [shard_range(H, r, world)[1] - shard_range(H, r, world)[0] for r in range(world)]
    if isinstance(part, tuple):
        return tuple(gather_rows(p.contiguous(), sizes, group) for p in part)
    return gather_rows(part.contiguous(), sizes, group)


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
