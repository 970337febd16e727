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
from ._lib import require as _require


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


def _eval_model_occupied(model, pts, view_dirs, density_grid, grid_res, on_device: bool = False):
    """``_eval_model`` on the samples in occupied cells only; the others get sigma = 0 (what the mask gives them) and
    rgb = 0 (their weight is 0 either way).  ``on_device``: the number of kept samples is not read back -- the model takes
    it from device memory and every tensor keeps its full size (hash-grid model only)."""
    R, N = pts.shape[0], pts.shape[1]
    if on_device:
        if not hasattr(model, "_tables"):
            raise NotImplementedError("skip_empty='device' is built for the hash-grid model; the 8x256 kernels take their "
                                      "point count from the host")
        count, slot, pts_k, dirs_k = ops.occupancy_compact(pts, density_grid, grid_res, view_dirs, N, read_count=False)
        rgb_k, sigma_k = model(pts_k, dirs_k, count=count)
        rgb, sigma = ops.expand_kept(rgb_k, sigma_k, None, slot)
        return rgb.view(R, N, 3), sigma.view(R, N)
    kept, slot, pts_k, dirs_k = ops.occupancy_compact(pts, density_grid, grid_res, view_dirs, N)
    if hasattr(model, "_params") and hasattr(model, "_skip_h_first"):
        rgb_k, sigma_k = ops.vanilla_mlp(pts_k, dirs_k, model._params(), model._skip_h_first)
    else:
        rgb_k, sigma_k = model(pts_k, dirs_k)
    rgb, sigma = ops.expand_kept(rgb_k, sigma_k, kept, slot)
    return rgb.view(R, N, 3), sigma.view(R, N)


def _linspace(near: float, far: float, n: int, dev) -> torch.Tensor:
    """torch.linspace(near, far, n) exactly as the reference's CPU call computes it (src/Camera.py:130): the
    GPU kernel of linspace rounds differently in places, and the depths are held to bit equality."""
    return torch.linspace(near, far, n).to(dev)


def render_rays(model, rays_o, rays_d, near: float, far: float, n_samples: int, perturb: bool = True,
                t_rand: Optional[torch.Tensor] = None, white_bkgd: bool = False, fine_model=None,
                n_importance: int = 0, u_fine: Optional[torch.Tensor] = None, ndc=None,
                density_grid: Optional[torch.Tensor] = None, grid_res: Optional[int] = None,
                z_lin: Optional[torch.Tensor] = None, samples=None, skip_empty=False):
    """Render a batch of rays [R,3].

    Returns a dict with ``rgb`` [R,3], ``weights`` [R,N], ``z_vals`` [R,N],
    ``depth`` [R] of the coarse pass and, when ``fine_model`` is given,
    ``rgb_fine``, ``weights_fine``, ``z_fine`` (the merged, sorted depths) and
    ``depth_fine``.  ``t_rand`` [R,N] / ``u_fine`` [R,n_importance] inject the
    uniform draws (drawn on the GPU when omitted).

    ``ndc = (H, W, focal, near_plane)`` renders a forward-facing scene in normalised device coordinates
    (NeRF 2020, appendix C; not in the reference): the rays are moved to the near plane and projected,
    ``near``/``far`` are then NDC depths (0 and 1), the view directions fed to the model stay the
    normalised world-space directions, and the compositor keeps ``Camera.volume_render``'s convention
    of distances in depth units (not scaled by the length of the NDC direction).
    ``density_grid`` (uint8, from ``get_density_grid``) zeroes the densities of samples in empty cells
    before compositing, as the hash-grid training render does (src/train_nerf_ngp.py:113-117).  With
    ``skip_empty`` those samples are dropped BEFORE the model instead (it runs on the occupied ones only): the same
    colours, weights and depths -- a sample with zero density has zero weight and no gradient -- for a fraction of the
    work; the sizes become data-dependent, so such a step cannot be captured into a HIP graph.  ``skip_empty="device"``
    (hash-grid model) keeps the sizes fixed instead: the number of kept samples stays in device memory, the kernels are
    launched for all of them and leave early -- nothing is read back and the step can be captured.
    ``samples = (z_vals [R,N], pts [R,N,3])`` supplies coarse samples already drawn (by the fused
    ``ops.raygen_sample``), in which case ``t_rand`` / ``perturb`` / ``z_lin`` are not used.
    """
    R = rays_o.shape[0]
    dev = rays_o.device
    view_dirs = ops.normalize_dirs(rays_d)   # rays_d / ||rays_d||, as render_image does (src/train_nerf_llff.py:42)
    if ndc is not None:
        H, W, focal, near_plane = ndc
        rays_o, rays_d = ops.ndc_rays(int(H), int(W), float(focal), float(near_plane), rays_o, rays_d)
    if samples is not None:
        _require(ndc is None, "samples drawn along world-space rays do not fit an NDC render")
        z, pts = samples
    else:
        if z_lin is None:
            z_lin = _linspace(near, far, n_samples, dev)
        if perturb and t_rand is None:
            t_rand = torch.rand(R, n_samples, device=dev)
        z, pts = ops.stratified_sample(rays_o, rays_d, z_lin, t_rand if perturb else None)
    skip = bool(skip_empty) and density_grid is not None
    on_device = skip_empty == "device"
    if skip:
        rgb, sigma = _eval_model_occupied(model, pts, view_dirs, density_grid, grid_res, on_device)
    else:
        rgb, sigma = _eval_model(model, pts, view_dirs, n_samples)
        if density_grid is not None:
            sigma = ops.occupancy_mask(sigma, pts, density_grid, grid_res)
    c_map, w, depth = ops.composite(rgb, sigma, z, white_bkgd)
    out = {"rgb": c_map, "weights": w, "z_vals": z, "depth": depth}
    if fine_model is not None and n_importance > 0:
        if u_fine is None:
            u_fine = torch.rand(R, n_importance, device=dev)
        _, _, z_all, pts_f = ops.pdf_sample(rays_o, rays_d, z, w, u_fine, want_fine=False)
        if skip:
            rgb_f, sigma_f = _eval_model_occupied(fine_model, pts_f, view_dirs, density_grid, grid_res, on_device)
        else:
            rgb_f, sigma_f = _eval_model(fine_model, pts_f, view_dirs, n_samples + n_importance)
            if density_grid is not None:
                sigma_f = ops.occupancy_mask(sigma_f, pts_f, density_grid, grid_res)
        f_map, w_f, depth_f = ops.composite(rgb_f, sigma_f, z_all, white_bkgd)
        out.update({"rgb_fine": f_map, "weights_fine": w_f, "z_fine": z_all, "depth_fine": depth_f})
    return out


def workspace_bytes_per_ray(model, n_samples: int, fine_model=None, n_importance: int = 0) -> int:
    """Bytes of activation workspace a training render keeps per ray (forward workspaces of both passes
    plus the larger backward scratch)."""
    from . import _lib as L
    lib = L.lib()

    def per_point(m):
        if hasattr(m, "_tables"):
            return lib.nerf_mlp_ngp_ws_floats_per_point(), lib.nerf_mlp_ngp_ws2_floats_per_point()
        return lib.nerf_mlp_vanilla_ws_floats_per_point(), lib.nerf_mlp_vanilla_ws2_floats_per_point()

    ws, ws2 = per_point(model)
    floats, scratch = ws * n_samples, ws2 * n_samples
    if fine_model is not None and n_importance > 0:
        ws_f, ws2_f = per_point(fine_model)
        floats += ws_f * (n_samples + n_importance)
        scratch = max(scratch, ws2_f * (n_samples + n_importance))
    return 4 * (floats + scratch)


def train_step(model, rays_o, rays_d, target, near: float, far: float, n_samples: int, *, loss_fn=None,
               chunk_rays: Optional[int] = None, workspace_budget: int = 32 << 30, buckets=None,
               grad_scale: Optional[float] = None, t_rand: Optional[torch.Tensor] = None,
               u_fine: Optional[torch.Tensor] = None, samples=None, z_lin: Optional[torch.Tensor] = None,
               **render_kw) -> torch.Tensor:
    """Forward + backward of one training step over R rays, in ray chunks, gradients accumulated.

    The reference bounds memory with a chunk loop over points inside ``render_image``
    (src/train_nerf_llff.py:50-56) while autograd still keeps every chunk's activations; here a chunk of
    rays runs forward AND backward before the next one starts, so the activation workspace (20 KB per
    point evaluation of the 8x256 MLP) is bounded by ``chunk_rays`` whatever R is -- 65,536 rays x (64 + 192)
    evaluations would need 300 GB otherwise.  ``chunk_rays`` defaults to the largest power of two whose
    workspace fits ``workspace_budget`` bytes.

    ``loss_fn(out, target_chunk)`` must be a mean over rays (default: ``nerf_loss``).  ``buckets`` (a
    ``nerf_amd.dist.GradBuckets``) receives the gradients in its flat buffers and overlaps the cross-rank
    sum with the last chunk's backward; without it the gradients accumulate into ``p.grad`` (zero them
    first).  The optimiser step is the caller's.  Returns the detached loss (0-d tensor).
    ``z_lin`` [n_samples]: the unperturbed depths, already on the GPU (default: computed on the host as the
    reference does and copied over -- a step that is to be captured into a HIP graph must pass them).
    """
    from .dist import accumulate_step
    R = rays_o.shape[0]
    loss_fn = loss_fn or nerf_loss
    if chunk_rays is None:
        per_ray = workspace_bytes_per_ray(model, n_samples, render_kw.get("fine_model"), render_kw.get("n_importance", 0))
        chunk_rays = 1 << max(0, (workspace_budget // max(per_ray, 1)).bit_length() - 1)
    chunk_rays = max(1, min(int(chunk_rays), R))
    if z_lin is None and samples is None:
        z_lin = _linspace(near, far, n_samples, rays_o.device)

    def loss_of_chunk(lo, hi):
        out = render_rays(model, rays_o[lo:hi], rays_d[lo:hi], near, far, n_samples,
                          t_rand=None if t_rand is None else t_rand[lo:hi],
                          u_fine=None if u_fine is None else u_fine[lo:hi], z_lin=z_lin,
                          samples=None if samples is None else (samples[0][lo:hi], samples[1][lo:hi]), **render_kw)
        return loss_fn(out, target[lo:hi])

    return accumulate_step(R, chunk_rays, loss_of_chunk, buckets, grad_scale)


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
                 n_importance: int = 0, rays_per_tile: int = 65536, return_depth: bool = False, ndc=None,
                 skip_empty: bool = False):
    """Full-image inference: rays [H,W,3] -> colour map [H,W,3] (and depth [H,W]).

    The counterpart of ``render_from_model`` (src/utils/analyze_capture_results.py:250-296): unperturbed
    samples, no autograd, sigma zeroed outside the occupancy grid when one is given.  The reference
    materialises all H*W*N points and loops over 16384-point chunks of them; here the loop is over tiles
    of rays and each tile runs sample -> model -> composite end to end.  With ``fine_model`` a second,
    importance-sampled pass follows (deterministic: evenly spaced CDF positions).  ``ndc`` and ``skip_empty`` as
    in ``render_rays``.
    """
    H, W = rays_o.shape[0], rays_o.shape[1]
    o_all, d_all = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    R = o_all.shape[0]
    dev = o_all.device
    out = torch.empty(R, 3, device=dev)
    depth = torch.empty(R, device=dev) if return_depth else None
    z_lin = _linspace(near, far, n_samples, dev)
    with torch.no_grad():
        for r0 in range(0, R, rays_per_tile):
            o, d = o_all[r0:r0 + rays_per_tile], d_all[r0:r0 + rays_per_tile]
            view_dirs = ops.normalize_dirs(d)
            if ndc is not None:
                o, d = ops.ndc_rays(int(ndc[0]), int(ndc[1]), float(ndc[2]), float(ndc[3]), o, d)
            z, pts = ops.stratified_sample(o, d, z_lin, None)
            skip = skip_empty and density_grid is not None
            if skip:
                rgb, sigma = _eval_model_occupied(model, pts, view_dirs, density_grid, grid_res)
            else:
                rgb, sigma = _eval_model(model, pts, view_dirs, n_samples)
                if density_grid is not None:
                    sigma = ops.occupancy_mask(sigma, pts, density_grid, grid_res)
            c_map, w, dep = ops.composite(rgb, sigma, z, white_bkgd)
            if fine_model is not None and n_importance > 0:
                u = torch.linspace(0.0, 1.0, n_importance, device=dev).expand(o.shape[0], n_importance).contiguous()
                _, _, z, pts = ops.pdf_sample(o, d, z, w, u, want_fine=False)
                if skip:
                    rgb, sigma = _eval_model_occupied(fine_model, pts, view_dirs, density_grid, grid_res)
                else:
                    rgb, sigma = _eval_model(fine_model, pts, view_dirs, n_samples + n_importance)
                    if density_grid is not None:
                        sigma = ops.occupancy_mask(sigma, pts, density_grid, grid_res)
                c_map, w, dep = ops.composite(rgb, sigma, z, white_bkgd)
            out[r0:r0 + rays_per_tile] = c_map
            if return_depth:
                depth[r0:r0 + rays_per_tile] = dep
    img = out.view(H, W, 3)
    return (img, depth.view(H, W)) if return_depth else img


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


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0, win_size: int = 7) -> torch.Tensor:
    """Mean structural similarity of two images [H,W,C] as the per-view log of the synthetic-scene loop computes it
    (src/train_nerf.py:174: scikit-image's ``structural_similarity(gt, img, channel_axis=2, data_range=1.0)`` with its
    defaults): a ``win_size`` x ``win_size`` box window, sample covariances (n / (n - 1)), K1 = 0.01, K2 = 0.03,
    averaged over the window positions that lie wholly inside the image and then over the channels (Wang et al. 2004).
    Evaluated in float64 on the tensors' device; a 0-d tensor comes back."""
    if x.shape != y.shape or x.dim() != 3:
        raise ValueError("ssim: two images [H,W,C] of one shape, got %s and %s" % (tuple(x.shape), tuple(y.shape)))
    if min(x.shape[0], x.shape[1]) < win_size:
        raise ValueError("ssim: the image is smaller than the %d-pixel window" % win_size)
    a, b = x.permute(2, 0, 1)[None].double(), y.permute(2, 0, 1)[None].double()
    box = lambda t: torch.nn.functional.avg_pool2d(t, win_size, stride=1)   # valid positions only
    n = win_size * win_size
    ua, ub = box(a), box(b)
    va, vb, vab = (n / (n - 1.0)) * (box(a * a) - ua * ua), (n / (n - 1.0)) * (box(b * b) - ub * ub), (n / (n - 1.0)) * (box(a * b) - ua * ub)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua * ua + ub * ub + c1) * (va + vb + c2))
    return s.mean(dim=(0, 2, 3)).mean()


def nerf_loss(out: dict, target: torch.Tensor) -> torch.Tensor:
    """MSE of the coarse colour (src/train_nerf_llff.py:116) plus, when a fine pass was rendered, the MSE of the fine
    colour: one fused op (``ops.render_loss``; GPU tensors only, like every op of this package)."""
    return ops.render_loss(out["rgb"], target, out.get("rgb_fine"))


def ngp_loss(out: dict, target: torch.Tensor, reg: float = 0.05) -> torch.Tensor:
    """The hash-grid loop's loss (src/train_nerf_ngp.py:221-222): MSE plus ``reg`` times the mean deviation of
    the rays' summed compositing weights from 1, as one fused op (``ops.render_loss``)."""
    return ops.render_loss(out["rgb"], target, weights=out["weights"], reg=reg)
