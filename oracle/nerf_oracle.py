"""CPU oracle for the NeRF render/train hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain CPU PyTorch, the arithmetic of the reference's hot
path so that the HIP kernels can be checked against it on a machine that does
not hold the reference.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``nerf_amd`` never does.

Pinning: every function that has a counterpart in the reference is checked
against golden vectors produced by importing the reference itself
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).  The reference has no hierarchical sampler,
no NDC transform and no depth output: ``sample_pdf``, ``ndc_rays`` and
``depth`` below follow the published NeRF algorithm (Mildenhall et al. 2020,
section 5.2 and appendix C) and are PARITY UNPINNED with respect to the
reference -- the HIP kernels are pinned to this file for those three.

Each function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Camera / rays                                         (src/Camera.py)
# --------------------------------------------------------------------------


def look_at_camera(eye: torch.Tensor, target: torch.Tensor,
                   up: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c2w [3,4] with columns (right, true_up, forward, eye).

    Follows src/Camera.py:6-30: forward = normalize(target-eye); the up
    vector defaults to +Y and is swapped for +Z when |forward.up| > 0.999;
    right = normalize(up x forward); true_up = forward x right.
    """
    eye = eye.float()
    target = target.float()
    fwd = target - eye
    fwd = fwd / torch.norm(fwd)
    if up is None:
        up = torch.tensor([0.0, 1.0, 0.0], dtype=eye.dtype)
    up = up.float()
    if torch.abs(torch.dot(fwd, up)) > 0.999:
        up = torch.tensor([0.0, 0.0, 1.0], dtype=eye.dtype)
    right = torch.cross(up, fwd, dim=0)
    right = right / torch.norm(right)
    true_up = torch.cross(fwd, right, dim=0)
    return torch.cat([torch.stack([right, true_up, fwd], dim=1), eye.view(3, 1)], dim=1)


def get_rays(c2w: torch.Tensor, focal: Sequence[float], H: int, W: int,
             pp: Optional[Sequence[float]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-pixel rays, src/Camera.py:49-72.

    Pixel (row j, col i) gets camera-space direction
    ((i-cx)/fx, -(j-cy)/fy, -1), rotated by c2w[:3,:3] and normalised; the
    origin is the translation column.  No half-pixel offset.
    """
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                          torch.arange(W, dtype=torch.float32), indexing="ij")
    if pp is not None:
        cx, cy = pp
    else:
        cx, cy = W * 0.5, H * 0.5
    d_cam = torch.stack([(i - cx) / focal[0], -(j - cy) / focal[1], -torch.ones_like(i)], dim=-1)
    rays_d = torch.sum(d_cam[..., None, :] * c2w[:3, :3], dim=-1)
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:3, 3].expand_as(rays_d)
    return rays_o, rays_d


def stratified_z(near: float, far: float, n_samples: int, shape: Tuple[int, ...],
                 t_rand: Optional[torch.Tensor]) -> torch.Tensor:
    """Depths along each ray, src/Camera.py:130-138.

    ``t_rand`` are the uniforms the reference would draw with
    ``torch.rand_like`` (None = no perturbation).  The reference's trailing
    sort is kept.
    """
    z = torch.linspace(near, far, n_samples).view(*([1] * len(shape)), n_samples).expand(*shape, n_samples)
    if t_rand is None:
        return z
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], dim=-1)
    lower = torch.cat([z[..., :1], mids], dim=-1)
    z = lower + (upper - lower) * t_rand
    z, _ = torch.sort(z, dim=-1)
    return z


def sample_points_along_rays(rays_o, rays_d, near, far, n_samples,
                             t_rand: Optional[torch.Tensor] = None):
    """pts = o + d*z, src/Camera.py:127-141 (uniforms injected)."""
    z = stratified_z(near, far, n_samples, tuple(rays_o.shape[:-1]), t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    return pts, z


def volume_render(rgb, sigma, z_vals, white_bkgd: bool = False, tail: float = 1e5):
    """Alpha compositing, src/Camera.py:143-183.

    delta_i = z_{i+1}-z_i with a 1e5 tail, alpha = 1-exp(-sigma*delta),
    T_i = prod_{j<i}(1-alpha_j+1e-10), w = alpha*T, C = sum w c
    (+ 1 - sum w when white_bkgd).  The older free function (src/utils/get_rays.py:198-236)
    is the same with tail = 1e10 and only the colour returned.
    """
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], tail)], dim=-1)
    alpha = 1.0 - torch.exp(-sigma * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha + 1e-10], dim=-1),
                          dim=-1)[..., :-1]
    weights = alpha * trans
    out = torch.sum(weights[..., None] * rgb, dim=-2)
    if white_bkgd:
        out = out + (1.0 - weights.sum(dim=-1, keepdim=True))
    return out, weights


# The older free functions of src/utils/get_rays.py (another camera convention).


def legacy_compute_stable_up(eye, target):
    """src/utils/get_rays.py:6-19."""
    fwd = target - eye
    fwd = fwd / torch.norm(fwd)
    guess = torch.tensor([0.0, 1.0, 0.0])
    if torch.allclose(fwd, guess, atol=1e-2) or torch.allclose(fwd, -guess, atol=1e-2):
        guess = torch.tensor([0.0, 0.0, 1.0])
    proj = guess - torch.dot(guess, fwd) * fwd
    return proj / torch.norm(proj)


def legacy_look_at_camera(eye, target, up):
    """src/utils/get_rays.py:151-170: right = forward x up, true_up = right x forward."""
    fwd = target - eye
    fwd = fwd / torch.norm(fwd)
    right = torch.linalg.cross(fwd, up)
    right = right / torch.norm(right)
    true_up = torch.linalg.cross(right, fwd)
    return torch.cat([torch.stack([right, true_up, fwd], dim=1), eye.view(3, 1)], dim=1)


def legacy_get_rays(H: int, W: int, focal: float, c2w: torch.Tensor):
    """src/utils/get_rays.py:115-147: [W,H,3] outputs (the meshgrid runs over the width first), camera
    looking along +z, principal point at the image centre."""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
    d_cam = torch.stack([(i - W * 0.5) / focal, -(j - H * 0.5) / focal, torch.ones_like(i)], dim=-1)
    rays_d = torch.sum(d_cam[..., None, :] * c2w[:3, :3], dim=-1)
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    return c2w[:3, 3].expand_as(rays_d), rays_d


def legacy_volume_render(rgb, sigma, z_vals):
    """src/utils/get_rays.py:198-236 (without its prints)."""
    return volume_render(rgb, sigma, z_vals, tail=1e10)[0]


def analytic_field_sphere(pts):
    """src/utils/get_rays.py:67-86: red unit ball of density 50."""
    rgb = torch.zeros_like(pts)
    rgb[..., 0] = 1.0
    return rgb, (torch.norm(pts, dim=-1) < 1.0).to(pts.dtype) * 50.0


def analytic_field_torus(pts):
    """src/utils/get_rays.py:88-112: hard red torus in the xz plane."""
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    rgb = torch.zeros_like(pts)
    rgb[..., 0] = 1.0
    return rgb, (((torch.sqrt(x ** 2 + z ** 2) - 1.2) ** 2 + y ** 2) < 0.3 ** 2).float() * 5.0


def depth(weights, z_vals):
    """Expected depth sum_i w_i z_i.  Not in the reference (its volume_render
    returns colour and weights only, src/Camera.py:183): parity unpinned."""
    return torch.sum(weights * z_vals, dim=-1)


# --------------------------------------------------------------------------
# Encoders                    (src/utils/positional_encoding.py, hash_grid_encoder.py)
# --------------------------------------------------------------------------


def positional_encoding(x: torch.Tensor, num_freqs: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)];
    src/utils/positional_encoding.py:14,30-36 (log-scale bands, no pi)."""
    bands = 2.0 ** torch.linspace(0, num_freqs - 1, num_freqs)
    parts = [x]
    for f in bands:
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, dim=-1)


HASH_PRIMES = (1, 2654435761, 805459861)  # src/utils/hash_grid_encoder.py:37


def hash_resolutions(n_levels=16, base_resolution=16, finest_resolution=512):
    """Per-level grid resolution, src/utils/hash_grid_encoder.py:24-25
    (float64 growth factor, truncation with int())."""
    b = np.exp((np.log(finest_resolution) - np.log(base_resolution)) / (n_levels - 1))
    return [int(base_resolution * (b ** i)) for i in range(n_levels)]


def hash_table_sizes(resolutions, log2_hashmap_size=19):
    """min(2^T, res^3) rows per level, src/utils/hash_grid_encoder.py:32."""
    return [min(2 ** log2_hashmap_size, r ** 3) for r in resolutions]


def hash_corner_indices(x: torch.Tensor, res: int, table_size: int):
    """Corner rows and interpolation weights for one level.

    Follows src/utils/hash_grid_encoder.py:56-82: pos = x*res, p0 = floor,
    w = pos - p0 taken before wrapping, both corners wrapped with a floor
    modulo, row = ((cx*1 + cy*2654435761 + cz*805459861) & 0xFFFFFFFF) % size.
    Returns idx int64 [P,8] in the corner order 000,001,010,011,100,101,110,111
    (digits are the x,y,z offsets) and w float32 [P,3].
    """
    pos = x * res
    p0 = pos.floor().long()
    w = pos - p0.float()
    p1 = (p0 + 1) % res
    p0 = p0 % res
    primes = torch.tensor(HASH_PRIMES, dtype=torch.long)
    cols = []
    for bx in (0, 1):
        for by in (0, 1):
            for bz in (0, 1):
                c = torch.stack([(p1 if bx else p0)[..., 0], (p1 if by else p0)[..., 1],
                                 (p1 if bz else p0)[..., 2]], dim=-1)
                cols.append(((c * primes).sum(dim=-1) & 0xFFFFFFFF) % table_size)
    return torch.stack(cols, dim=-1), w


def hash_encode(x: torch.Tensor, tables: Sequence[torch.Tensor], resolutions: Sequence[int]):
    """Multiresolution hash encoding [P, 2*levels], src/utils/hash_grid_encoder.py:42-107.
    Interpolation runs along x, then y, then z."""
    feats = []
    for tab, res in zip(tables, resolutions):
        idx, w = hash_corner_indices(x, res, tab.shape[0])
        f = [tab[idx[:, c]] for c in range(8)]  # 000,001,010,011,100,101,110,111
        wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
        fx00 = f[0] * (1 - wx) + f[4] * wx
        fx01 = f[1] * (1 - wx) + f[5] * wx
        fx10 = f[2] * (1 - wx) + f[6] * wx
        fx11 = f[3] * (1 - wx) + f[7] * wx
        fxy0 = fx00 * (1 - wy) + fx10 * wy
        fxy1 = fx01 * (1 - wy) + fx11 * wy
        feats.append(fxy0 * (1 - wz) + fxy1 * wz)
    return torch.cat(feats, dim=-1)


# --------------------------------------------------------------------------
# Models.  Parameters are created in the reference's order with the
# reference's initialisers, so torch.manual_seed(s) gives identical tensors.
# --------------------------------------------------------------------------


class NeRFOracle(nn.Module):
    """src/models/NeRF.py:16-109.  Skip input at layer 4 is [x_enc | h]."""

    def __init__(self, xyz_freqs=10, dir_freqs=4, hidden_dim=256, num_layers=8, skips=(4,)):
        super().__init__()
        self.xyz_freqs, self.dir_freqs, self.skips = xyz_freqs, dir_freqs, tuple(skips)
        in_xyz = 3 * (2 * xyz_freqs + 1)
        in_dir = 3 * (2 * dir_freqs + 1)
        self.layers_xyz = nn.ModuleList()
        for i in range(num_layers):
            k = in_xyz if i == 0 else (in_xyz + hidden_dim if i in self.skips else hidden_dim)
            self.layers_xyz.append(nn.Linear(k, hidden_dim))
        self.sigma_out = nn.Linear(hidden_dim, 1)
        self.feature_layer = nn.Linear(hidden_dim, hidden_dim)
        self.rgb_layers = nn.Sequential(nn.Linear(hidden_dim + in_dir, hidden_dim // 2), nn.ReLU(),
                                        nn.Linear(hidden_dim // 2, 3), nn.Sigmoid())

    def forward(self, x, view_dirs, masks=None, record=None):
        """``masks`` (bool/0-1 [P, 8*256 + 128], trunk layers then the colour hidden layer) replaces every
        ReLU by a multiplication with the given pattern: the network becomes the piecewise-linear branch
        another implementation took, so gradients can be compared without ReLUs whose pre-activation is
        within rounding of zero flipping.  ``record``: a list that receives every pre-activation."""
        xe = positional_encoding(x, self.xyz_freqs)
        h = xe
        for i, lin in enumerate(self.layers_xyz):
            if i in self.skips:
                h = torch.cat([xe, h], dim=-1)
            pre = lin(h)
            if record is not None:
                record.append(pre.detach())
            h = F.relu(pre) if masks is None else pre * masks[:, 256 * i:256 * (i + 1)].to(pre.dtype)
        sigma = F.softplus(self.sigma_out(h))
        if view_dirs is None:
            return torch.zeros(x.shape[0], 3), sigma
        de = positional_encoding(view_dirs, self.dir_freqs)
        if masks is None and record is None:
            rgb = self.rgb_layers(torch.cat([self.feature_layer(h), de], dim=-1))
        else:
            pre = self.rgb_layers[0](torch.cat([self.feature_layer(h), de], dim=-1))
            if record is not None:
                record.append(pre.detach())
            hr = F.relu(pre) if masks is None else pre * masks[:, 2048:2176].to(pre.dtype)
            rgb = torch.sigmoid(self.rgb_layers[2](hr))
        return rgb, sigma


class FastMLPNeRFOracle(nn.Module):
    """src/models/FastMLPNeRF.py:7-70.  Skip input at layer 4 is [h | x_enc]."""

    def __init__(self, num_layers=8, hidden_dim=256, skips=(4,), num_encoding_fn_xyz=10,
                 num_encoding_fn_dir=4):
        super().__init__()
        self.lx, self.ld = num_encoding_fn_xyz, num_encoding_fn_dir
        in_xyz = 3 * (2 * self.lx + 1)
        in_dir = 3 * (2 * self.ld + 1)
        self.mlp_xyz = nn.ModuleList()
        for i in range(num_layers):
            k = in_xyz if i == 0 else (hidden_dim + in_xyz if i in skips else hidden_dim)
            self.mlp_xyz.append(nn.Linear(k, hidden_dim))
        self.fc_sigma = nn.Linear(hidden_dim, 1)
        self.fc_feature = nn.Linear(hidden_dim, hidden_dim)
        self.mlp_rgb = nn.Sequential(nn.Linear(hidden_dim + in_dir, hidden_dim // 2), nn.ReLU(),
                                     nn.Linear(hidden_dim // 2, 3), nn.Sigmoid())

    def forward(self, x, view_dirs):
        xe = positional_encoding(x, self.lx)
        de = positional_encoding(view_dirs, self.ld)
        h = xe
        for i, lin in enumerate(self.mlp_xyz):
            if i == 4:
                h = torch.cat([h, xe], dim=-1)
            h = F.relu(lin(h))
        sigma = F.softplus(self.fc_sigma(h))
        rgb = self.mlp_rgb(torch.cat([self.fc_feature(h), de], dim=-1))
        return rgb, sigma


class HashGridOracle(nn.Module):
    """src/utils/hash_grid_encoder.py:6-37 (tables as nn.Embedding so the state
    dict keys and the RNG consumption match: default normal init, then uniform)."""

    def __init__(self, n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                 base_resolution=16, finest_resolution=512):
        super().__init__()
        self.resolutions = hash_resolutions(n_levels, base_resolution, finest_resolution)
        self.hash_size = 2 ** log2_hashmap_size
        self.embeddings = nn.ModuleList()
        for r in self.resolutions:
            emb = nn.Embedding(min(self.hash_size, r ** 3), n_features_per_level)
            nn.init.uniform_(emb.weight, a=1e-4, b=1e-3)
            self.embeddings.append(emb)
        self.register_buffer("primes", torch.tensor(HASH_PRIMES, dtype=torch.long))

    def forward(self, x):
        return hash_encode(x, [e.weight for e in self.embeddings], self.resolutions)


class FastNGPNeRFOracle(nn.Module):
    """src/models/FastNGPNeRF.py:30-124."""

    def __init__(self, n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                 base_resolution=16, finest_resolution=512, hidden_dim=64):
        super().__init__()
        self.hash_encoder = HashGridOracle(n_levels, n_features_per_level, log2_hashmap_size,
                                           base_resolution, finest_resolution)
        self.fc_density = nn.Sequential(nn.Linear(n_levels * n_features_per_level, hidden_dim),
                                        nn.ReLU(), nn.Linear(hidden_dim, 16))
        for m in self.fc_density:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=1e-4)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.fc_density[2].bias, 1.5)
        self.fc_color = nn.Sequential(nn.Linear(15 + 27, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 3))
        for m in self.fc_color:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=1e-4)
                nn.init.constant_(m.bias, 0)
        self.skip_proj = nn.Linear(15, hidden_dim, bias=False)

    def forward(self, x, view_dirs=None):
        raw = self.fc_density(self.hash_encoder(x))
        sigma = F.softplus(raw[..., 0] + 1.5, beta=1)
        feat = raw[..., 1:]
        if view_dirs is not None:
            view_dirs = view_dirs / view_dirs.norm(dim=-1, keepdim=True)
            de = positional_encoding(view_dirs, 4)
        else:
            de = torch.zeros(x.shape[0], 27)
        h = torch.relu(self.fc_color[0](torch.cat([feat, de], dim=-1)) + self.skip_proj(feat))
        return torch.sigmoid(self.fc_color[2](h)), sigma

    @torch.no_grad()
    def query_density(self, x):
        return F.softplus(self.fc_density(self.hash_encoder(x))[:, 0] + 1.5, beta=1)


# --------------------------------------------------------------------------
# Build-defined stages (no reference code): hierarchical sampling and NDC.
# --------------------------------------------------------------------------


def wave_inclusive_scan(v: torch.Tensor) -> torch.Tensor:
    """Inclusive prefix sum along the last axis in the fixed association order
    of a log-step (Hillis-Steele) scan: step d = 1,2,4,... adds element i-d to
    element i.  The HIP kernel performs the same float32 additions with
    wavefront shuffles, so the results are bit-identical."""
    c = v.clone()
    n = c.shape[-1]
    d = 1
    while d < n:
        nxt = c.clone()
        nxt[..., d:] = c[..., d:] + c[..., :-d]
        c = nxt
        d *= 2
    return c


def sample_pdf(z_vals: torch.Tensor, weights: torch.Tensor, u: torch.Tensor):
    """Inverse-CDF sampling of the coarse weights (NeRF 2020, section 5.2).

    z_vals, weights: [R,N] from the coarse pass; u: [R,Nf] in [0,1).
    bins = midpoints of z (N-1), w = weights[1:-1] + 1e-5 (N-2),
    cdf = [0, scan(w)/scan(w)[-1]] (N-1), idx = #(cdf <= u) (searchsorted
    right), below = max(idx-1,0), above = min(idx,N-2), t = (u-cdf[below]) /
    (cdf[above]-cdf[below]) with a denominator below 1e-5 replaced by 1,
    z_f = bins[below] + t*(bins[above]-bins[below]).
    Returns z_fine [R,Nf], idx int32 [R,Nf], z_all = sort(z_vals U z_fine).
    Parity unpinned with respect to the reference (it has no such stage).
    """
    bins = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    w = weights[..., 1:-1] + 1e-5
    c = wave_inclusive_scan(w)
    cdf = torch.cat([torch.zeros_like(c[..., :1]), c / c[..., -1:]], dim=-1)
    idx = torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True)
    below = torch.clamp(idx - 1, min=0)
    above = torch.clamp(idx, max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b_lo, b_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c_hi - c_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c_lo) / denom
    z_fine = b_lo + t * (b_hi - b_lo)
    z_all, _ = torch.sort(torch.cat([z_vals, z_fine], dim=-1), dim=-1)
    return z_fine, idx.to(torch.int32), z_all


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o, rays_d):
    """Forward-facing NDC transform (NeRF 2020, appendix C): shift origins to
    the z = -near plane, then project.  Parity unpinned (not in the reference)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    ax, ay = -focal / (0.5 * W), -focal / (0.5 * H)
    o0 = ax * (o[..., 0] / o[..., 2])
    o1 = ay * (o[..., 1] / o[..., 2])
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = ax * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = ay * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2.0 * near / o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------
# Render glue and the training step (the op sequence the CPU baseline times)
# --------------------------------------------------------------------------


def run_model_chunked(model, pts, dirs, chunk, masks=None):
    """Chunk loop of src/train_nerf_llff.py:31-64 / src/train_nerf_ngp.py:80-126."""
    rgbs, sigmas = [], []
    for i in range(0, pts.shape[0], chunk):
        if masks is None:
            r, s = model(pts[i:i + chunk], dirs[i:i + chunk])
        else:
            r, s = model(pts[i:i + chunk], dirs[i:i + chunk], masks=masks[i:i + chunk])
        rgbs.append(r)
        sigmas.append(s)
    return torch.cat(rgbs, 0), torch.cat(sigmas, 0)


def xyz_to_idx(x, y, z, res: int = 32):
    """src/utils/analyze_capture_results.py:230-231 (callers never pass res)."""
    return x + y * res + z * res * res


def occupancy_mask(sigma, pts, density_grid, grid_res: int):
    """sigma with the samples in empty occupancy cells zeroed, src/train_nerf_ngp.py:113-117 (torch.where in
    place of the in-place write: same values, same zero gradient).  A cell index past the grid, an
    IndexError in the reference, is clamped (the HIP kernel does the same)."""
    cell = 1.0 / grid_res
    x, y, z = ((pts.reshape(-1, 3) + 1) / 2 / cell).long().T.unbind(0)
    idx = xyz_to_idx(x, y, z).clamp(0, density_grid.numel() - 1)
    keep = (density_grid[idx] == 1).view(sigma.shape)
    return torch.where(keep, sigma, torch.zeros_like(sigma))


def get_density_grid(model, grid_res: int):
    """src/utils/analyze_capture_results.py:233-248: lattice points of [-1,1]^3 whose density exceeds 1."""
    r = torch.arange(grid_res)
    coords = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), dim=-1).view(-1, 3)
    grid = torch.zeros(grid_res ** 3, dtype=torch.uint8)
    with torch.no_grad():
        for i in range(0, coords.shape[0], 65536):
            c = coords[i:i + 65536]
            mask = model.query_density((c.float() / (grid_res - 1)) * 2 - 1) > 1.0
            grid[xyz_to_idx(c[:, 0], c[:, 1], c[:, 2])] |= mask.to(torch.uint8)
    return grid


def render_rays(model, rays_o, rays_d, near, far, n_samples, t_rand=None, chunk=2048,
                white_bkgd=False, fine_model=None, u_fine=None, ndc=None, density_grid=None, grid_res=None,
                masks=None, masks_fine=None, z_fixed=None):
    """rays [R,3] -> colour [R,3] (and the fine-pass colour when fine_model is
    given).  Coarse pass = the reference's render_image; the fine pass is the
    build-defined hierarchical stage.  ndc = (H, W, focal, near_plane): forward-facing NDC rays
    (view directions stay the normalised world directions; distances stay depth differences).
    density_grid: occupancy masking of sigma before compositing (src/train_nerf_ngp.py:113-117).
    masks / masks_fine: forced ReLU patterns (NeRFOracle.forward); z_fixed: the merged fine-pass depths to use
    instead of this function's own inverse-CDF result (so that both sides evaluate the same points)."""
    R = rays_o.shape[0]
    vd = rays_d / rays_d.norm(dim=-1, keepdim=True)
    if ndc is not None:
        rays_o, rays_d = ndc_rays(int(ndc[0]), int(ndc[1]), float(ndc[2]), float(ndc[3]), rays_o, rays_d)
    pts, z = sample_points_along_rays(rays_o, rays_d, near, far, n_samples, t_rand)
    dirs = vd[:, None, :].expand_as(pts).reshape(-1, 3)
    rgb, sigma = run_model_chunked(model, pts.reshape(-1, 3), dirs, chunk, masks)
    sigma = sigma.view(R, n_samples)
    if density_grid is not None:
        sigma = occupancy_mask(sigma, pts, density_grid, grid_res)
    c_map, w = volume_render(rgb.view(R, n_samples, 3), sigma, z, white_bkgd)
    if fine_model is None:
        return c_map, w, z, None
    z_all = sample_pdf(z, w.detach(), u_fine)[2] if z_fixed is None else z_fixed
    nf = z_all.shape[-1]
    pts_f = rays_o[:, None, :] + rays_d[:, None, :] * z_all[..., None]
    dirs_f = vd[:, None, :].expand_as(pts_f).reshape(-1, 3)
    rgb_f, sigma_f = run_model_chunked(fine_model, pts_f.reshape(-1, 3), dirs_f, chunk, masks_fine)
    sigma_f = sigma_f.view(R, nf)
    if density_grid is not None:
        sigma_f = occupancy_mask(sigma_f, pts_f, density_grid, grid_res)
    f_map, w_f = volume_render(rgb_f.view(R, nf, 3), sigma_f, z_all, white_bkgd)
    return c_map, w, z, (f_map, w_f, z_all)


def psnr(x, y):
    """-10 log10(mse), src/train_nerf_llff.py:18-19."""
    return -10.0 * torch.log10(torch.mean((x - y) ** 2))


def analytic_field(pts):
    """Torus + sphere radiance field, src/utils/get_rays.py:25-64; the only
    data-free ground-truth scene the reference contains."""
    x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
    ring = torch.sqrt(x ** 2 + y ** 2)
    torus = torch.exp(-((torch.sqrt((ring - 1.0) ** 2 + z ** 2) - 0.3) ** 2) * 40.0)
    ball = torch.exp(-((torch.sqrt(x ** 2 + y ** 2 + z ** 2) - 0.4) ** 2) * 60.0)
    sigma = torch.max(torus, ball)
    red = torch.tensor([1.0, 0.2, 0.2])
    blue = torch.tensor([0.2, 0.2, 1.0])
    rgb = torch.where((torus > ball)[..., None], red, blue).expand_as(pts)
    return rgb, sigma
