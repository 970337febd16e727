"""Drop-in for the older free-function helpers of src/utils/get_rays.py: same names,
arguments and return shapes, with ray generation, sampling and compositing running as the
gfx950 HIP kernels that also back ``Camera`` (tensors must live on the GPU).

These helpers predate ``Camera`` and follow another convention, which is kept:
``get_rays`` shoots along +z of the pose and returns ``[W, H, 3]`` (its meshgrid runs over the
width first), ``sample_points_along_rays`` never perturbs, and ``volume_render`` appends a
distance of 1e10 (not 1e5) after the last sample and returns the colour only.
"""
from typing import Tuple

import torch

from .. import ops


def compute_stable_up(eye, target):
    """An up vector orthogonal to the viewing direction (src/utils/get_rays.py:6-19): +y, or +z when
    the view is within 1e-2 of vertical, with its component along the view removed.  Host-side."""
    fwd = target - eye
    fwd = fwd / torch.norm(fwd)
    guess = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float32, device=fwd.device)
    if torch.allclose(fwd, guess, atol=1e-2) or torch.allclose(fwd, -guess, atol=1e-2):
        guess = torch.tensor([0.0, 0.0, 1.0], device=fwd.device)
    proj = guess - torch.dot(guess, fwd) * fwd
    return proj / torch.norm(proj)


def look_at_camera(eye, target, up):
    """c2w [3,4] = [right | true_up | forward | eye] with right = forward x up and
    true_up = right x forward (src/utils/get_rays.py:151-170; ``Camera.look_at_camera`` uses the
    opposite handedness).  Host-side 3-vector algebra."""
    fwd = target - eye
    fwd = fwd / torch.norm(fwd)
    right = torch.linalg.cross(fwd, up)
    right = right / torch.norm(right)
    true_up = torch.linalg.cross(right, fwd)
    return torch.cat([torch.stack([right, true_up, fwd], dim=1), eye.view(3, 1)], dim=1)


def get_rays(H, W, focal, c2w):
    """rays_o, rays_d ``[W, H, 3]`` (src/utils/get_rays.py:115-147): pixel (i, j), i along the width, looks
    along ((i - W/2)/focal, -(j - H/2)/focal, +1) in the camera frame.

    Runs on ``nerf_raygen`` (whose camera looks along -z) with the pose's third column negated -- the same
    products and sums, bit for bit -- and the pixels requested in width-major order."""
    H, W = int(H), int(W)
    pose = c2w[:3, :4].float().clone()
    pose[:, 2] = -pose[:, 2]
    dev = pose.device
    i = torch.arange(W, device=dev, dtype=torch.int32)
    j = torch.arange(H, device=dev, dtype=torch.int32)
    pix = (j[None, :] * W + i[:, None]).reshape(-1)          # entry (i, j) of the [W, H] grid is pixel j*W + i
    f = float(focal)
    o, d = ops.raygen(pose, f, f, W * 0.5, H * 0.5, H, W, pix)
    return o.view(W, H, 3), d.view(W, H, 3)


def sample_points_along_rays(rays_o, rays_d, near, far, N_samples):
    """pts [A,B,N,3], z_vals [A,B,N]: evenly spaced depths, no perturbation
    (src/utils/get_rays.py:176-196).  Two leading dimensions, as in the reference."""
    A, B = rays_o.shape[:2]
    z_lin = torch.linspace(near, far, N_samples).to(rays_o.device)   # the reference's CPU linspace values
    z, pts = ops.stratified_sample(rays_o.reshape(A * B, 3), rays_d.reshape(A * B, 3), z_lin, None)
    return pts.view(A, B, N_samples, 3), z.view(A, B, N_samples)


def volume_render(rgb, sigma, z_vals, verbose: bool = False):
    """final_rgb [A,B,3] (src/utils/get_rays.py:198-236): ``Camera.volume_render`` with a tail distance of
    1e10 and the colour as the only return value.  The reference prints four statistics on every call
    (each a device-to-host sync); here only with ``verbose=True``, and only the two that the returned
    tensor allows."""
    A, B, N = sigma.shape
    out, _, _ = ops.composite(rgb, sigma.to(rgb.device), z_vals.to(rgb.device), False, tail=ops.TAIL_LEGACY)
    if verbose:
        print("Image min:", out.min().item(), "max:", out.max().item())
    return out


# ---- the analytic radiance fields the reference trains and tests against (src/utils/get_rays.py:25-112) ----
# Plain tensor formulas (any device); they produce training targets, they are not on the kernel path.

def evaluate_fake_radiance_field_complex(pts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Red torus (radii 1.0 / 0.3, xy plane) around a blue sphere shell (radius 0.4): soft densities
    exp(-40 d_torus^2) and exp(-60 d_sphere^2), united by max (src/utils/get_rays.py:25-64)."""
    x, y, z = pts.unbind(-1)
    d_torus = torch.sqrt((torch.sqrt(x ** 2 + y ** 2) - 1.0) ** 2 + z ** 2) - 0.3
    d_sphere = torch.sqrt(x ** 2 + y ** 2 + z ** 2) - 0.4
    s_torus = torch.exp(-(d_torus ** 2) * 40.0)
    s_sphere = torch.exp(-(d_sphere ** 2) * 60.0)
    colours = torch.tensor([[0.2, 0.2, 1.0], [1.0, 0.2, 0.2]], device=pts.device)
    rgb = colours[(s_torus > s_sphere).long()].expand_as(pts)
    return rgb, torch.max(s_torus, s_sphere)


def evaluate_fake_radiance_field_sphere(pts):
    """Solid red unit ball of density 50 (src/utils/get_rays.py:67-86)."""
    inside = torch.norm(pts, dim=-1) < 1.0
    rgb = torch.zeros_like(pts)
    rgb[..., 0] = 1.0
    return rgb, inside.to(pts.dtype) * 50.0


def evaluate_fake_radiance_field_torus(pts):
    """Hard red torus in the xz plane, radii 1.2 / 0.3, density 5 (src/utils/get_rays.py:88-112)."""
    x, y, z = pts.unbind(-1)
    inside = (torch.sqrt(x ** 2 + z ** 2) - 1.2) ** 2 + y ** 2 < 0.3 ** 2
    rgb = torch.zeros_like(pts)
    rgb[..., 0] = 1.0
    return rgb, inside.float() * 5.0
