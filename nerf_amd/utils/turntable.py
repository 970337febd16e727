"""Novel-view sweeps around a captured object: the pose helpers and the frame loop of the reference's result
viewers (src/utils/analyze_capture_results.py:25-40, 92-115, 298-346; src/utils/analyze_results.py:78-94), with every
frame rendered by the tiled device renderer.  Frames come back as one uint8 tensor; writing a GIF is left to the
caller (``PIL.Image.fromarray(frames[k].cpu().numpy())``)."""
import math
from typing import Optional, Tuple

import torch

from ..Camera import Camera
from ..render import render_image


def estimate_pivot(poses: torch.Tensor) -> torch.Tensor:
    """The point the cameras look at: least-squares intersection of their optical axes
    (src/utils/analyze_capture_results.py:25-40).  poses [N,3,4] (or [N,4,4]) camera-to-world -> [3]."""
    centres = poses[:, :3, 3]
    forward = -poses[:, :3, 2]
    # distance^2 of p from the line (c, f) is |(I - f f^T)(p - c)|^2: the normal equations summed over the cameras
    proj = torch.eye(3, dtype=poses.dtype, device=poses.device) - forward[:, :, None] * forward[:, None, :]
    rhs = (proj @ centres[:, :, None]).squeeze(-1)
    return torch.linalg.lstsq(proj.sum(0), rhs.sum(0)).solution


def turntable_pose(c2w_start: torch.Tensor, pivot: torch.Tensor, t: float) -> torch.Tensor:
    """c2w_start [3,4] turned by 360 degrees * t about the world +Y axis through ``pivot``: position and local
    frame rotate together (src/utils/analyze_capture_results.py:92-115).  Returns [3,4]."""
    axis = torch.tensor([0.0, 1.0, 0.0], device=c2w_start.device, dtype=c2w_start.dtype)
    rot = Camera.axis_angle_to_matrix(axis, torch.tensor(2.0 * math.pi * t, device=c2w_start.device, dtype=c2w_start.dtype))
    eye = pivot + rot @ (c2w_start[:3, 3] - pivot)
    return torch.cat([rot @ c2w_start[:3, :3], eye[:, None]], dim=-1)


def to_uint8(img: torch.Tensor) -> torch.Tensor:
    """clamp(0, 1) * 255 truncated to bytes, as the viewers write their frames."""
    return (img.clamp(0.0, 1.0) * 255.0).to(torch.uint8)


def render_turntable(model, c2w_start: torch.Tensor, pivot: torch.Tensor, H: int, W: int, focal: Tuple[float, float],
                     pp: Optional[Tuple[float, float]] = None, num_frames: int = 120, near: float = 0.05, far: float = 1.2,
                     n_samples: int = 64, grid_res: Optional[int] = None, density_grid: Optional[torch.Tensor] = None,
                     **render_kw):
    """The frame loop of ``render_one_turntable`` (src/utils/analyze_capture_results.py:324-346): ``num_frames``
    poses at t = k / num_frames, rays from each, one full-image inference per frame.  Returns (frames uint8
    [F,H,W,3] on the model's device, poses [F,3,4]).  ``render_kw`` goes to ``render_image`` (``skip_empty``,
    ``fine_model`` ...)."""
    frames, poses = [], []
    for k in range(num_frames):
        c2w = turntable_pose(c2w_start, pivot, k / num_frames)
        cam = Camera(eye=c2w[:, 3], target=None, c2w=c2w, H=H, W=W, focal=focal, pp=pp)
        rays_o, rays_d = cam.get_rays()
        img = render_image(model, rays_o, rays_d, near, far, n_samples=n_samples, grid_res=grid_res,
                           density_grid=density_grid, **render_kw)
        frames.append(to_uint8(img))
        poses.append(c2w)
    return torch.stack(frames), torch.stack(poses)


def render_model_image(model, cam: Camera, n_samples: int, near: float = 2.0, far: float = 6.0, **render_kw):
    """One preview image of ``cam`` as bytes [H,W,3] on the host (src/utils/analyze_results.py:78-94).  The reference
    leaves the sampler's jitter on here; a preview does not need it and this one is deterministic."""
    dev = next(model.parameters()).device   # the reference forms the rays on the host and moves them; here they are formed there
    on_dev = Camera(eye=cam.eye, target=cam.target, focal=cam.focal, H=cam.H, W=cam.W, c2w=cam.c2w.to(dev), pp=cam.principal_point)
    rays_o, rays_d = on_dev.get_rays()
    return to_uint8(render_image(model, rays_o, rays_d, near, far, n_samples=n_samples, **render_kw)).cpu().numpy()
