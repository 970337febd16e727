"""Tensor-level wrappers and autograd Functions over the C ABI (libnerf_hip.so).

PyTorch is used here for device memory, streams and the autograd graph only;
all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib as L


# ---------------------------------------------------------------- rays ----
def raygen(c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, H: int, W: int,
           pix: Optional[torch.Tensor] = None):
    """Rays of an H x W pinhole camera (all pixels, or the int32 pixel ids in ``pix``)."""
    L.need_gpu(c2w, pix)
    pose = L.f32c(c2w[:3, :4])
    R = H * W if pix is None else pix.numel()
    if pix is not None:
        pix = pix.to(torch.int32).contiguous()
    o = torch.empty(R, 3, device=pose.device, dtype=torch.float32)
    d = torch.empty_like(o)
    L.check(L.lib().nerf_raygen(L.ptr(pose), fx, fy, cx, cy, H, W, L.ptr(pix), R, L.ptr(o), L.ptr(d),
                                L.stream_of(pose)), "nerf_raygen")
    return o, d


def raygen_multi(cams: torch.Tensor, H: int, W: int, cam_id: torch.Tensor, pix: torch.Tensor,
                 images: Optional[torch.Tensor] = None):
    """Rays of a batch drawn across cameras.  cams [n,16] (3x4 pose rows, fx, fy, cx, cy); ray r is pixel
    ``pix[r]`` of camera ``cam_id[r]``; with ``images`` [n,H,W,3] the pixels' colours come back as well."""
    L.need_gpu(cams, cam_id, pix, images)
    cams = L.f32c(cams)
    L.require(cams.dim() == 2 and cams.shape[1] == 16, f"cams must be [n,16], got {tuple(cams.shape)}")
    cam_id = cam_id.to(torch.int32).contiguous()
    pix = pix.to(torch.int32).contiguous()
    R = pix.numel()
    L.require(cam_id.numel() == R, "one camera id per pixel id")
    if images is not None:
        images = L.f32c(images)
        L.require(tuple(images.shape) == (cams.shape[0], H, W, 3), f"images must be [{cams.shape[0]},{H},{W},3]")
    o = torch.empty(R, 3, device=cams.device, dtype=torch.float32)
    d = torch.empty_like(o)
    rgb = torch.empty_like(o) if images is not None else None
    L.check(L.lib().nerf_raygen_multi(L.ptr(cams), cams.shape[0], H, W, L.ptr(cam_id), L.ptr(pix), R, L.ptr(images),
                                      L.ptr(o), L.ptr(d), L.ptr(rgb), L.stream_of(cams)), "nerf_raygen_multi")
    return (o, d, rgb) if images is not None else (o, d)


def stratified_sample(rays_o: torch.Tensor, rays_d: torch.Tensor, z_lin: torch.Tensor,
                      u: Optional[torch.Tensor], want_pts: bool = True):
    """rays [R,3], base depths z_lin [N], uniforms u [R,N] or None -> z [R,N], pts [R,N,3]."""
    L.need_gpu(rays_o, rays_d, z_lin, u)
    o, d, zl = L.f32c(rays_o), L.f32c(rays_d), L.f32c(z_lin)
    R, N = o.shape[0], zl.numel()
    if u is not None:
        u = L.f32c(u)
        L.require(u.shape == (R, N), f"u must be [{R},{N}], got {tuple(u.shape)}")
    z = torch.empty(R, N, device=o.device, dtype=torch.float32)
    pts = torch.empty(R, N, 3, device=o.device, dtype=torch.float32) if want_pts else None
    L.check(L.lib().nerf_stratified_sample(L.ptr(o), L.ptr(d), L.ptr(zl), L.ptr(u), R, N, L.ptr(z), L.ptr(pts),
                                           L.stream_of(o)), "nerf_stratified_sample")
    return z, pts


def raygen_sample(c2w, fx, fy, cx, cy, H, W, pix, z_lin, u):
    """Fused ray generation + stratified sampling -> rays_o, rays_d [R,3], z [R,N], pts [R,N,3]."""
    L.need_gpu(c2w, pix, z_lin, u)
    pose = L.f32c(c2w[:3, :4])
    zl = L.f32c(z_lin)
    N = zl.numel()
    R = H * W if pix is None else pix.numel()
    if pix is not None:
        pix = pix.to(torch.int32).contiguous()
    if u is not None:
        u = L.f32c(u)
        L.require(u.shape == (R, N), f"u must be [{R},{N}], got {tuple(u.shape)}")
    dev = pose.device
    o = torch.empty(R, 3, device=dev)
    d = torch.empty(R, 3, device=dev)
    z = torch.empty(R, N, device=dev)
    pts = torch.empty(R, N, 3, device=dev)
    L.check(L.lib().nerf_raygen_sample(L.ptr(pose), fx, fy, cx, cy, H, W, L.ptr(pix), R, L.ptr(zl), L.ptr(u), N,
                                       L.ptr(o), L.ptr(d), L.ptr(z), L.ptr(pts), L.stream_of(pose)),
            "nerf_raygen_sample")
    return o, d, z, pts


def pdf_sample(rays_o, rays_d, z, weights, u, want_idx: bool = False, want_fine: bool = True):
    """Inverse-CDF fine sampling -> z_fine [R,Nf] | None, idx int32 [R,Nf] | None, z_all [R,Nc+Nf], pts [R,Nc+Nf,3]."""
    L.need_gpu(rays_o, rays_d, z, weights, u)
    o, d, z, w, u = (L.f32c(t) for t in (rays_o, rays_d, z, weights.detach(), u))
    R, Nc = z.shape
    Nf = u.shape[1]
    dev = z.device
    zf = torch.empty(R, Nf, device=dev) if want_fine else None
    idx = torch.empty(R, Nf, device=dev, dtype=torch.int32) if want_idx else None
    z_all = torch.empty(R, Nc + Nf, device=dev)
    pts = torch.empty(R, Nc + Nf, 3, device=dev)
    L.check(L.lib().nerf_pdf_sample(L.ptr(o), L.ptr(d), L.ptr(z), L.ptr(w), L.ptr(u), R, Nc, Nf, L.ptr(zf), L.ptr(idx),
                                    L.ptr(z_all), L.ptr(pts), L.stream_of(z)), "nerf_pdf_sample")
    return zf, idx, z_all, pts


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o, rays_d):
    L.need_gpu(rays_o, rays_d)
    o, d = L.f32c(rays_o).reshape(-1, 3), L.f32c(rays_d).reshape(-1, 3)
    oo, dd = torch.empty_like(o), torch.empty_like(d)
    L.check(L.lib().nerf_ndc_rays(L.ptr(o), L.ptr(d), o.shape[0], H, W, focal, near, L.ptr(oo), L.ptr(dd),
                                  L.stream_of(o)), "nerf_ndc_rays")
    return oo.view(rays_o.shape), dd.view(rays_d.shape)


def normalize_dirs(rays_d):
    """rays_d / ||rays_d|| over the last axis, as one launch."""
    L.need_gpu(rays_d)
    d = L.f32c(rays_d.detach()).reshape(-1, 3)
    out = torch.empty_like(d)
    L.check(L.lib().nerf_normalize_dirs(L.ptr(d), d.shape[0], L.ptr(out), L.stream_of(d)), "nerf_normalize_dirs")
    return out.view(rays_d.shape)


# ------------------------------------------------------------ encoders ----
def encode_pe(x: torch.Tensor, num_freqs: int) -> torch.Tensor:
    """[..., D] -> [..., D*(2L+1)].  The encoder input carries no gradient in the
    reference's pipelines (sample points are constants), so none is produced."""
    L.need_gpu(x)
    xc = L.f32c(x.detach())
    D = xc.shape[-1]
    P = xc.numel() // D
    out = torch.empty(*xc.shape[:-1], D * (2 * num_freqs + 1), device=xc.device)
    L.check(L.lib().nerf_encode_pe(L.ptr(xc), P, D, num_freqs, L.ptr(out), L.stream_of(xc)), "nerf_encode_pe")
    return out


class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, sizes, *tables):
        L.need_gpu(x, *tables)
        xc = L.f32c(x.detach())
        P, levels = xc.shape[0], len(tables)
        tabs = [L.f32c(t.detach()) for t in tables]
        out = torch.empty(P, 2 * levels, device=xc.device)
        L.check(L.lib().nerf_hashgrid_fwd(L.ptr(xc), P, L.ptr_array(tabs), L.int_array(res), L.int_array(sizes), levels,
                                          L.ptr(out), None, None, L.stream_of(xc)), "nerf_hashgrid_fwd")
        ctx.save_for_backward(xc)
        ctx.meta = (tuple(res), tuple(sizes), [t.shape for t in tables])
        return out

    @staticmethod
    def backward(ctx, d_out):
        (xc,) = ctx.saved_tensors
        res, sizes, shapes = ctx.meta
        g = L.f32c(d_out)
        grads = [torch.zeros(s, device=xc.device) for s in shapes]
        work, nbytes = scatter_workspace(xc.shape[0], sizes, xc.device)
        L.check(L.lib().nerf_hashgrid_bwd(L.ptr(xc), xc.shape[0], L.ptr(g), L.int_array(res), L.int_array(sizes),
                                          len(res), L.ptr_array(grads), L.ptr(work), nbytes, L.stream_of(xc)),
                "nerf_hashgrid_bwd")
        return (None, None, None, *grads)


def scatter_workspace(P: int, sizes, device):
    """Scratch for the binned table-gradient scatter (None, 0 when the batch is too small to bin)."""
    nbytes = L.lib().nerf_hashgrid_bwd_workspace_bytes(P, L.int_array(sizes), len(sizes))
    if not nbytes:
        return None, 0
    return torch.empty(nbytes, device=device, dtype=torch.uint8), nbytes


def hashgrid_encode(x: torch.Tensor, tables: Sequence[torch.Tensor], res: Sequence[int]) -> torch.Tensor:
    sizes = [int(t.shape[0]) for t in tables]
    return _HashGrid.apply(x, list(res), sizes, *tables)


def hashgrid_debug(x: torch.Tensor, tables: Sequence[torch.Tensor], res: Sequence[int]):
    """Features plus the hashed row of every corner (int32 [levels,8,P]) and the
    interpolation weights ([levels,P,3]) -- for parity checks."""
    xc = L.f32c(x)
    P, levels = xc.shape[0], len(tables)
    tabs = [L.f32c(t.detach()) for t in tables]
    sizes = [int(t.shape[0]) for t in tabs]
    out = torch.empty(P, 2 * levels, device=xc.device)
    idx = torch.empty(levels, 8, P, device=xc.device, dtype=torch.int32)
    w = torch.empty(levels, P, 3, device=xc.device)
    L.check(L.lib().nerf_hashgrid_fwd(L.ptr(xc), P, L.ptr_array(tabs), L.int_array(res), L.int_array(sizes), levels,
                                      L.ptr(out), L.ptr(idx), L.ptr(w), L.stream_of(xc)), "nerf_hashgrid_fwd")
    return out, idx, w


# ----------------------------------------------------------- composite ----
class _Composite(torch.autograd.Function):
    """(rgb [R,N,3], sigma [R,N], z [R,N]) -> (colour [R,3], weights [R,N], depth [R])."""

    @staticmethod
    def forward(ctx, rgb, sigma, z, white_bkgd, tail):
        L.need_gpu(rgb, sigma, z)
        rgb_c, sig_c, z_c = L.f32c(rgb.detach()), L.f32c(sigma.detach()), L.f32c(z.detach())
        R, N = sig_c.shape
        out = torch.empty(R, 3, device=rgb_c.device)
        w = torch.empty(R, N, device=rgb_c.device)
        depth = torch.empty(R, device=rgb_c.device)
        L.check(L.lib().nerf_composite_fwd(L.ptr(rgb_c), L.ptr(sig_c), L.ptr(z_c), R, N, int(bool(white_bkgd)),
                                           float(tail), L.ptr(out), L.ptr(w), L.ptr(depth), L.stream_of(rgb_c)),
                "nerf_composite_fwd")
        ctx.save_for_backward(rgb_c, sig_c, z_c)
        ctx.white = int(bool(white_bkgd))
        ctx.tail = float(tail)
        ctx.set_materialize_grads(False)
        return out, w, depth

    @staticmethod
    def backward(ctx, d_out, d_w, d_depth):
        rgb_c, sig_c, z_c = ctx.saved_tensors
        R, N = sig_c.shape
        d_out = L.f32c(d_out) if d_out is not None else torch.zeros(R, 3, device=rgb_c.device)
        d_w = L.f32c(d_w) if d_w is not None else None
        d_depth = L.f32c(d_depth) if d_depth is not None else None
        d_rgb = torch.empty_like(rgb_c)
        d_sigma = torch.empty_like(sig_c)
        L.check(L.lib().nerf_composite_bwd(L.ptr(rgb_c), L.ptr(sig_c), L.ptr(z_c), L.ptr(d_out), L.ptr(d_w),
                                           L.ptr(d_depth), R, N, ctx.white, ctx.tail, L.ptr(d_rgb), L.ptr(d_sigma),
                                           L.stream_of(rgb_c)), "nerf_composite_bwd")
        return d_rgb, d_sigma, None, None, None


TAIL_CAMERA = 1e5    # Camera.volume_render, src/Camera.py:154
TAIL_LEGACY = 1e10   # volume_render of src/utils/get_rays.py:214


def composite(rgb, sigma, z, white_bkgd: bool = False, tail: float = TAIL_CAMERA):
    """Alpha compositing over the last sample axis; leading axes are flattened to rays.
    ``tail`` is the distance appended after the last sample."""
    lead = sigma.shape[:-1]
    N = sigma.shape[-1]
    out, w, depth = _Composite.apply(rgb.reshape(-1, N, 3), sigma.reshape(-1, N), z.reshape(-1, N), white_bkgd, tail)
    return out.view(*lead, 3), w.view(*lead, N), depth.view(*lead)


class _OccupancyMask(torch.autograd.Function):
    """sigma [P] -> sigma with the samples in empty occupancy cells zeroed; they get no gradient."""

    @staticmethod
    def forward(ctx, sigma, pts, grid, grid_res):
        L.need_gpu(sigma, pts, grid)
        sc, pc = L.f32c(sigma.detach()).reshape(-1), L.f32c(pts.detach()).reshape(-1, 3)
        L.require(pc.shape[0] == sc.shape[0], "one point per density")
        if grid.dtype != torch.uint8:
            raise L.NerfHipError("the occupancy grid must be uint8")
        gc = grid.contiguous()
        out = torch.empty_like(sc)
        L.check(L.lib().nerf_occupancy_mask(L.ptr(pc), L.ptr(sc), sc.shape[0], L.ptr(gc), gc.numel(), int(grid_res),
                                            L.ptr(out), L.stream_of(sc)), "nerf_occupancy_mask")
        ctx.save_for_backward(pc, gc)
        ctx.grid_res = int(grid_res)
        return out.view(sigma.shape)

    @staticmethod
    def backward(ctx, d_out):
        pc, gc = ctx.saved_tensors
        g = L.f32c(d_out).reshape(-1)
        d_in = torch.empty_like(g)
        L.check(L.lib().nerf_occupancy_mask(L.ptr(pc), L.ptr(g), g.shape[0], L.ptr(gc), gc.numel(), ctx.grid_res,
                                            L.ptr(d_in), L.stream_of(g)), "nerf_occupancy_mask")
        return d_in.view(d_out.shape), None, None, None


def occupancy_mask(sigma, pts, grid, grid_res: int):
    """Zero the densities of the samples ``pts`` that fall into empty cells of the uint8 occupancy ``grid``
    (src/train_nerf_ngp.py:113-117)."""
    return _OccupancyMask.apply(sigma, pts, grid, int(grid_res))


def occupancy_compact(pts, grid, grid_res: int, dirs=None, dirs_div: int = 1, read_count: bool = True):
    """The samples of ``pts`` [P,3] that lie in occupied cells of the uint8 ``grid``, in their own order:
    returns ``(kept int32 [K], slot int32 [P], pts_kept [K,3], dirs_kept [K,3] | None)`` -- ``kept`` the indices,
    ``slot[p]`` the position of sample p among them or -1, ``dirs_kept`` the rows of ``dirs`` (one per ``dirs_div``
    consecutive samples) the kept samples belong to.  K is read back from the GPU (one small device-to-host copy:
    a step whose tensors have data-dependent sizes cannot be captured into a HIP graph).
    ``read_count=False`` reads nothing back: ``(count int32 [1] on the GPU, slot, pts_kept [P,3], dirs_kept [P,3] | None)``
    with the arrays at their full size and only their first ``count`` rows meaningful -- for callees that take the
    count from device memory (``ngp_mlp(..., count=)``)."""
    L.need_gpu(pts, grid, dirs)
    pc = L.f32c(pts.detach()).reshape(-1, 3)
    if grid.dtype != torch.uint8:
        raise L.NerfHipError("the occupancy grid must be uint8")
    gc = grid.contiguous()
    dc = L.f32c(dirs.detach()) if dirs is not None else None
    P, dev = pc.shape[0], pc.device
    lib = L.lib()
    work = torch.empty(lib.nerf_occupancy_compact_workspace_ints(P), device=dev, dtype=torch.int32)
    kept = torch.empty(P, device=dev, dtype=torch.int32)
    slot = torch.empty(P, device=dev, dtype=torch.int32)
    pts_k = torch.empty(P, 3, device=dev)
    dirs_k = torch.empty(P, 3, device=dev) if dc is not None else None
    count = torch.empty(1, device=dev, dtype=torch.int32)
    if dc is not None:
        L.require(dc.shape == ((P + dirs_div - 1) // dirs_div, 3), "dirs must hold one row per dirs_div samples")
    L.check(lib.nerf_occupancy_compact(L.ptr(pc), P, L.ptr(gc), gc.numel(), int(grid_res), L.ptr(dc), int(dirs_div),
                                       L.ptr(work), L.ptr(kept), L.ptr(slot), L.ptr(pts_k), L.ptr(dirs_k), L.ptr(count),
                                       L.stream_of(pc)), "nerf_occupancy_compact")
    if not read_count:
        return count, slot, pts_k, dirs_k
    K = int(count.item())
    return kept[:K], slot, pts_k[:K], (dirs_k[:K] if dirs_k is not None else None)


class _ExpandKept(torch.autograd.Function):
    """(rgb_kept [K,3], sigma_kept [K]) -> (rgb [P,3], sigma [P]): the kept samples' values at their places, zeros at the
    dropped ones; backward gathers the kept samples' gradients.  ``kept`` None: the count stayed on the device, the
    compact arrays have all P rows and ``slot`` alone says which of them are in use."""

    @staticmethod
    def forward(ctx, rgb_k, sigma_k, kept, slot):
        L.need_gpu(rgb_k, sigma_k, kept, slot)
        rc, sc = L.f32c(rgb_k.detach()).reshape(-1, 3), L.f32c(sigma_k.detach()).reshape(-1)
        P = slot.shape[0]
        K = kept.shape[0] if kept is not None else P
        L.require(rc.shape[0] == K and sc.shape[0] == K, "one model output per kept sample")
        rgb = torch.empty(P, 3, device=slot.device)
        sigma = torch.empty(P, device=slot.device)
        L.check(L.lib().nerf_expand_kept(L.ptr(rc), L.ptr(sc), L.ptr(slot), P, L.ptr(rgb), L.ptr(sigma), L.stream_of(slot)),
                "nerf_expand_kept")
        ctx.by_slot = kept is None
        ctx.save_for_backward(slot if kept is None else kept)
        ctx.shapes = (rgb_k.shape, sigma_k.shape)
        ctx.set_materialize_grads(False)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        (index,) = ctx.saved_tensors
        K = index.shape[0]
        d_rgb = L.f32c(d_rgb) if d_rgb is not None else None
        d_sigma = L.f32c(d_sigma) if d_sigma is not None else None
        g_rgb = torch.empty(K, 3, device=index.device)
        g_sigma = torch.empty(K, device=index.device)
        if ctx.by_slot:   # rows behind the count stay as they are: their consumer does not read them
            L.check(L.lib().nerf_gather_kept_by_slot(L.ptr(d_rgb), L.ptr(d_sigma), L.ptr(index), K, L.ptr(g_rgb), L.ptr(g_sigma),
                                                     L.stream_of(index)), "nerf_gather_kept_by_slot")
        else:
            L.check(L.lib().nerf_gather_kept(L.ptr(d_rgb), L.ptr(d_sigma), L.ptr(index), K, L.ptr(g_rgb), L.ptr(g_sigma),
                                             L.stream_of(index)), "nerf_gather_kept")
        return g_rgb.view(ctx.shapes[0]), g_sigma.view(ctx.shapes[1]), None, None


def expand_kept(rgb_kept, sigma_kept, kept, slot):
    return _ExpandKept.apply(rgb_kept, sigma_kept, kept, slot)


class _RenderLoss(torch.autograd.Function):
    """loss = mse(rgb, target) [+ mse(rgb_fine, target)] [+ reg * mean |1 - weights.sum(-1)|] in two launches,
    its gradient in one (src/train_nerf_llff.py:116, src/train_nerf_ngp.py:221-222)."""

    @staticmethod
    def forward(ctx, rgb, target, rgb_fine, weights, reg):
        L.need_gpu(rgb, target)
        lib = L.lib()
        c, t = L.f32c(rgb.detach()), L.f32c(target.detach())
        f = L.f32c(rgb_fine.detach()) if rgb_fine is not None else None
        w = L.f32c(weights.detach()) if weights is not None else None
        R = c.shape[0]
        L.require(c.shape == (R, 3) and t.shape == (R, 3) and R > 0, "rgb and target must be [R,3]")
        L.require(f is None or f.shape == (R, 3), "rgb_fine must be [R,3]")
        L.require(w is None or (w.dim() == 2 and w.shape[0] == R), "weights must be [R,N]")
        N = int(w.shape[1]) if w is not None else 0
        work = torch.empty(lib.nerf_render_loss_workspace_floats(R), device=c.device)
        loss = torch.empty((), device=c.device)
        L.check(lib.nerf_render_loss_fwd(L.ptr(c), L.ptr(f), L.ptr(t), L.ptr(w), R, N, float(reg), L.ptr(work), L.ptr(loss),
                                         L.stream_of(c)), "nerf_render_loss_fwd")
        ctx.save_for_backward(c, t, f, w)
        ctx.meta = (R, N, float(reg), rgb.shape, None if rgb_fine is None else rgb_fine.shape,
                    None if weights is None else weights.shape)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        c, t, f, w = ctx.saved_tensors
        R, N, reg, s_rgb, s_fine, s_w = ctx.meta
        need_c, _, need_f, need_w, _ = ctx.needs_input_grad
        g = L.f32c(d_loss).reshape(1)
        d_c = torch.empty_like(c) if need_c else None
        d_f = torch.empty_like(f) if (need_f and f is not None) else None
        d_w = torch.empty_like(w) if (need_w and w is not None) else None
        L.check(L.lib().nerf_render_loss_bwd(L.ptr(c), L.ptr(f), L.ptr(t), L.ptr(w), R, N, reg, L.ptr(g), L.ptr(d_c), L.ptr(d_f),
                                             L.ptr(d_w), L.stream_of(c)), "nerf_render_loss_bwd")
        return (None if d_c is None else d_c.view(s_rgb), None, None if d_f is None else d_f.view(s_fine),
                None if d_w is None else d_w.view(s_w), None)


def render_loss(rgb, target, rgb_fine=None, weights=None, reg: float = 0.0) -> torch.Tensor:
    """The training loss of the reference's loops as one fused op: ``mse_loss(rgb, target)`` (src/train_nerf_llff.py:116),
    plus ``mse_loss(rgb_fine, target)`` for a hierarchical render, plus ``reg * (1 - weights.sum(-1)).abs().mean()``
    (the hash-grid loop's regulariser, src/train_nerf_ngp.py:221-222, reg = 0.05).  The target receives no gradient."""
    return _RenderLoss.apply(rgb, target, rgb_fine, weights if reg != 0.0 else None, float(reg))


# ---------------------------------------------------------- fused MLPs ----
def _ceil32(n: int) -> int:
    return (n + 31) // 32 * 32


def vanilla_pack(params, skip_h_first: bool) -> torch.Tensor:
    """Re-order the 24 parameter tensors into the MFMA operand layout (one launch)."""
    lib = L.lib()
    ps = [L.f32c(p.detach()) for p in params]
    packed = torch.empty(lib.nerf_mlp_vanilla_pack_floats(), device=ps[0].device)
    L.check(lib.nerf_mlp_vanilla_pack(L.ptr_array(ps), int(skip_h_first), L.ptr(packed), L.stream_of(packed)),
            "nerf_mlp_vanilla_pack")
    return packed


def vanilla_fwd_raw(x, dirs, dirs_div: int, packed, save: bool):
    """x [P,3], dirs [ceil(P/dirs_div),3] -> rgb [P,3], sigma [P,1], workspace (or None)."""
    lib = L.lib()
    P = x.shape[0]
    dev = x.device
    rgb = torch.empty(P, 3, device=dev)
    sigma = torch.empty(P, 1, device=dev)
    ws = torch.empty(_ceil32(P) * lib.nerf_mlp_vanilla_ws_floats_per_point(), device=dev) if save and P else None
    L.check(lib.nerf_mlp_vanilla_fwd(L.ptr(x), L.ptr(dirs), int(dirs_div), P, L.ptr(packed), L.ptr(rgb), L.ptr(sigma),
                                     L.ptr(ws), L.stream_of(x)), "nerf_mlp_vanilla_fwd")
    return rgb, sigma, ws


def vanilla_bwd_raw(d_rgb, d_sigma, packed, rgb, sigma, ws, ws2, skip_h_first: bool, grads, phases: int = 3):
    """Accumulate the parameter gradients into ``grads`` (24 zeroed tensors)."""
    P = rgb.shape[0]
    L.check(L.lib().nerf_mlp_vanilla_bwd(L.ptr(d_rgb), L.ptr(d_sigma), P, L.ptr(packed), L.ptr(rgb), L.ptr(sigma),
                                         L.ptr(ws), L.ptr(ws2), int(skip_h_first), L.ptr_array(grads), int(phases),
                                         L.stream_of(rgb)), "nerf_mlp_vanilla_bwd")


def vanilla_ws2(P: int, device) -> torch.Tensor:
    return torch.empty(_ceil32(P) * L.lib().nerf_mlp_vanilla_ws2_floats_per_point(), device=device)


def zero_grads_like(shapes, device):
    """One zeroed flat buffer viewed as per-parameter gradients (a single memset)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    starts, off = [], 0
    for n in sizes:   # 256-byte aligned views: these tensors become p.grad, which the optimiser reads 16 bytes per lane
        starts.append(off)
        off += (n + 63) // 64 * 64
    flat = torch.zeros(off, device=device)
    return [flat[o:o + n].view(s) for s, n, o in zip(shapes, sizes, starts)]


# Where the parameter gradients of the fused MLPs go.  By default backward() returns them as new tensors and
# autograd adds them to p.grad: one zero-fill and one add per parameter and backward pass (for the hash-grid model
# 25 launches and 250 MB of traffic per step, an eighth of a 4096-ray step).  The kernels accumulate with atomics
# anyway, so while a sink is installed (grad_sink(...): accumulate_step does it for its GradBuckets) they
# accumulate straight into the tensors the sink names -- the buckets' p.grad views -- and autograd receives no
# gradient for those parameters; the sink is told afterwards, in place of the post-accumulate hooks.
# One sink per process at a time (backward runs on autograd's threads, so it cannot be thread-local): two training
# steps running concurrently in one process must not both install one.
_GRAD_SINK = None


class grad_sink:
    """``with grad_sink(sink):`` -- ``sink.targets(params)`` returns the tensors to accumulate the gradients of
    ``params`` into (same shapes, float32, contiguous, zeroed by the sink's owner when a step starts) or None to
    decline; ``sink.expect(params)`` is called from the forward pass of every op call that will deliver through the
    sink, ``sink.done(params)`` once that call's backward kernels are enqueued."""

    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        global _GRAD_SINK
        self.prev, _GRAD_SINK = _GRAD_SINK, self.sink
        return self

    def __exit__(self, *exc):
        global _GRAD_SINK
        _GRAD_SINK = self.prev
        return False


def _grad_expect(params):
    """Forward of a fused op in training mode: tell the sink, if it takes these parameters, that one backward
    call will deliver their gradients."""
    sink = _GRAD_SINK
    if sink is not None and sink.targets(params) is not None:
        sink.expect(params)
        return True
    return False


def _grad_targets(params, shapes, device, expected):
    """(tensors the backward kernels accumulate into, the sink that owns them or None).  Only a call whose forward
    announced itself to the sink delivers through it."""
    sink = _GRAD_SINK
    if expected and sink is not None:
        targets = sink.targets(params)
        if targets is not None:
            return targets, sink
    return zero_grads_like(shapes, device), None


class _VanillaMLP(torch.autograd.Function):
    """Fused 8x256 NeRF MLP: (x [P,3], dirs, 24 parameters) -> (rgb [P,3], sigma [P,1]).

    Parameters arrive in the order W0,b0,...,W7,b7,Wsigma,bsigma,Wfeat,bfeat,
    Wrgb0,brgb0,Wrgb2,brgb2.  Gradients are produced for the parameters only:
    the sample positions and view directions are constants in every pipeline
    of the reference.
    """

    @staticmethod
    def forward(ctx, x, dirs, dirs_div, skip_h_first, grad_mode, *params):
        L.require(len(params) == 24, "the 8x256 model has 24 parameter tensors")
        L.need_gpu(x, dirs, *params)
        xc, dc = L.f32c(x.detach()), L.f32c(dirs.detach())
        P = xc.shape[0]
        L.require(xc.shape == (P, 3), "x must be [N,3]")
        L.require(dc.shape == ((P + dirs_div - 1) // dirs_div, 3), "view_dirs must be [N,3] (or one row per ray)")
        packed = vanilla_pack(params, skip_h_first)
        # needs_input_grad mirrors requires_grad even under no_grad, and grad mode is always off in here:
        # the caller's grad mode arrives as an argument
        train = grad_mode and any(ctx.needs_input_grad[5:])
        rgb, sigma, ws = vanilla_fwd_raw(xc, dc, dirs_div, packed, train)
        if train:
            ctx.save_for_backward(packed, rgb, sigma, ws)
            ctx.meta = (P, bool(skip_h_first), [p.shape for p in params])
            ctx.params = params
            ctx.expected = P > 0 and _grad_expect(params)
        ctx.set_materialize_grads(False)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        packed, rgb, sigma, ws = ctx.saved_tensors
        P, skip, shapes = ctx.meta
        dev = rgb.device
        if P == 0:
            return (None, None, None, None, None, *zero_grads_like(shapes, dev))
        grads, sink = _grad_targets(ctx.params, shapes, dev, ctx.expected)
        d_rgb = L.f32c(d_rgb) if d_rgb is not None else torch.zeros(P, 3, device=dev)
        d_sigma = L.f32c(d_sigma) if d_sigma is not None else torch.zeros(P, 1, device=dev)
        vanilla_bwd_raw(d_rgb, d_sigma, packed, rgb, sigma, ws, vanilla_ws2(P, dev), skip, grads)
        if sink is not None:
            sink.done(ctx.params)
            return (None,) * (5 + len(grads))
        return (None, None, None, None, None, *grads)


def vanilla_mlp(x, dirs, params, skip_h_first: bool, dirs_div: int = 1):
    return _VanillaMLP.apply(x, dirs, int(dirs_div), bool(skip_h_first), torch.is_grad_enabled(), *params)


def vanilla_relu_masks(ws: torch.Tensor, P: int) -> torch.Tensor:
    """uint8 [P, 2176]: which ReLUs of the forward pass that filled ``ws`` were open (8 x 256 trunk neurons,
    then the 128 of the colour hidden layer) -- for parity checks."""
    out = torch.empty(P, 2176, device=ws.device, dtype=torch.uint8)
    L.check(L.lib().nerf_mlp_vanilla_relu_masks(L.ptr(ws), P, L.ptr(out), L.stream_of(ws)), "nerf_mlp_vanilla_relu_masks")
    return out


class _NgpMLP(torch.autograd.Function):
    """Fused hash-grid NeRF: (x [P,3], dirs | None, 16 tables, 9 MLP parameters) ->
    (rgb [P,3], sigma [P]).  MLP parameter order: fc_density.0.{weight,bias},
    fc_density.2.{weight,bias}, fc_color.0.{weight,bias}, fc_color.2.{weight,bias},
    skip_proj.weight.  Gradients go to the tables and the parameters.
    ``count`` (one int32 on the GPU, or None): only the first ``count`` points are in use -- x, dirs and the outputs
    keep their full size P, nothing is read back, rows behind the count are neither read nor written."""

    @staticmethod
    def forward(ctx, x, dirs, dirs_div, res, grad_mode, count, *tensors):
        tables, params = tensors[:16], tensors[16:]
        L.require(len(tables) == 16 and len(params) == 9, "the hash-grid model has 16 tables and 9 other parameter tensors")
        L.need_gpu(x, dirs, *tensors)
        lib = L.lib()
        xc = L.f32c(x.detach())
        dc = L.f32c(dirs.detach()) if dirs is not None else None
        P = xc.shape[0]
        L.require(xc.shape == (P, 3), "x must be [N,3]")
        dev = xc.device
        st = L.stream_of(xc)
        tabs = [L.f32c(t.detach()) for t in tables]
        ps = [L.f32c(p.detach()) for p in params]
        sizes = [int(t.shape[0]) for t in tabs]
        packed = torch.empty(lib.nerf_mlp_ngp_pack_floats(), device=dev)
        L.check(lib.nerf_mlp_ngp_pack(L.ptr_array(ps), L.ptr(packed), st), "nerf_mlp_ngp_pack")
        rgb = torch.empty(P, 3, device=dev)
        sigma = torch.empty(P, device=dev)
        train = grad_mode and any(ctx.needs_input_grad[6:])   # see _VanillaMLP.forward
        ws = torch.empty(_ceil32(P) * lib.nerf_mlp_ngp_ws_floats_per_point(), device=dev) if train and P else None
        enc = torch.empty(_ceil32(P) * 32, device=dev)   # the encoding, level-major, between the two launches
        if count is None:
            L.check(lib.nerf_mlp_ngp_fwd(L.ptr(xc), L.ptr(dc), int(dirs_div), P, L.ptr_array(tabs), L.int_array(res),
                                         L.int_array(sizes), L.ptr(packed), L.ptr(rgb), L.ptr(sigma), L.ptr(ws), L.ptr(enc), st),
                    "nerf_mlp_ngp_fwd")
        else:
            L.need_gpu(count)
            L.require(count.dtype == torch.int32 and count.numel() == 1, "count must be one int32")
            L.check(lib.nerf_mlp_ngp_fwd_counted(L.ptr(xc), L.ptr(dc), int(dirs_div), P, L.ptr(count), L.ptr_array(tabs),
                                                 L.int_array(res), L.int_array(sizes), L.ptr(packed), L.ptr(rgb), L.ptr(sigma),
                                                 L.ptr(ws), L.ptr(enc), st), "nerf_mlp_ngp_fwd_counted")
        ctx.count = count
        if train:
            ctx.save_for_backward(xc, packed, rgb, sigma, ws)
            ctx.meta = (P, list(res), sizes, [t.shape for t in tensors])
            ctx.params = tensors
            ctx.expected = P > 0 and _grad_expect(tensors)
        ctx.set_materialize_grads(False)
        return rgb, sigma

    @staticmethod
    def backward(ctx, d_rgb, d_sigma):
        xc, packed, rgb, sigma, ws = ctx.saved_tensors
        P, res, sizes, shapes = ctx.meta
        lib = L.lib()
        dev = rgb.device
        count = ctx.count
        if P == 0:
            return (None, None, None, None, None, None, *zero_grads_like(shapes, dev))
        grads, sink = _grad_targets(ctx.params, shapes, dev, ctx.expected)
        d_rgb = L.f32c(d_rgb) if d_rgb is not None else torch.zeros(P, 3, device=dev)
        d_sigma = L.f32c(d_sigma) if d_sigma is not None else torch.zeros(P, device=dev)
        ws2 = torch.empty(_ceil32(P) * lib.nerf_mlp_ngp_ws2_floats_per_point(), device=dev)
        work, nbytes = scatter_workspace(P, sizes, dev)

        def run(phases):
            if count is None:
                L.check(lib.nerf_mlp_ngp_bwd(L.ptr(xc), L.ptr(d_rgb), L.ptr(d_sigma), P, L.int_array(res), L.int_array(sizes),
                                             L.ptr(packed), L.ptr(rgb), L.ptr(sigma), L.ptr(ws), L.ptr(ws2),
                                             L.ptr_array(grads[:16]), L.ptr_array(grads[16:]), L.ptr(work), nbytes, phases,
                                             L.stream_of(rgb)), "nerf_mlp_ngp_bwd")
            else:
                L.check(lib.nerf_mlp_ngp_bwd_counted(L.ptr(xc), L.ptr(d_rgb), L.ptr(d_sigma), P, L.ptr(count), L.int_array(res),
                                                     L.int_array(sizes), L.ptr(packed), L.ptr(rgb), L.ptr(sigma), L.ptr(ws),
                                                     L.ptr(ws2), L.ptr_array(grads[:16]), L.ptr_array(grads[16:]), L.ptr(work),
                                                     nbytes, phases, L.stream_of(rgb)), "nerf_mlp_ngp_bwd_counted")

        if sink is not None:
            # the table gradients (41.7 MB, the one sizeable exchange of data-parallel training) are complete after the
            # first phase: reported now, their all-reduce runs under the weight-gradient kernels of the second
            run(1)
            sink.done(ctx.params[:16])
            run(2)
            sink.done(ctx.params[16:])
            return (None,) * (6 + len(grads))
        run(3)
        return (None, None, None, None, None, None, *grads)


def ngp_mlp(x, dirs, tables, params, res, dirs_div: int = 1, count=None):
    return _NgpMLP.apply(x, dirs, int(dirs_div), list(res), torch.is_grad_enabled(), count, *tables, *params)


# ------------------------------------------------------------ optimiser ----
def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ with the 2-norm (src/train_nerf.py:165) as three launches over all
    gradients and no host round trip.  Returns the total norm as a 0-d device tensor."""
    import ctypes as C
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    L.need_gpu(*grads)
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous():
            raise L.NerfHipError("clip_grad_norm_ needs contiguous float32 gradients")
    lib = L.lib()
    n = len(grads)
    numel = (C.c_int64 * n)(*[g.numel() for g in grads])
    dev = grads[0].device
    work = torch.empty(lib.nerf_clip_grad_norm_workspace_floats(n, numel), device=dev)
    total = torch.empty((), device=dev)
    L.check(lib.nerf_clip_grad_norm(n, L.ptr_array(grads), numel, float(max_norm), L.ptr(work), L.ptr(total),
                                    L.stream_of(grads[0])), "nerf_clip_grad_norm")
    return total
