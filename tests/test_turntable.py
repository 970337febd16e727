"""Pose helpers of the novel-view sweep (src/utils/analyze_capture_results.py:25-40, 92-115) against plain numpy
restatements, and -- on the GPU -- the frame loop against the oracle rendering the same poses."""
import math

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O


def _ring_poses(n, centre, radius, height):
    poses = []
    for k in range(n):
        a = 2 * math.pi * k / n
        eye = torch.tensor([centre[0] + radius * math.cos(a), centre[1] + height * (1 + 0.3 * math.sin(3 * a)),
                            centre[2] + radius * math.sin(a)])
        poses.append(O.look_at_camera(eye, torch.tensor(centre), torch.tensor([0.0, 1.0, 0.0])))
    return torch.stack(poses)


def test_estimate_pivot_recovers_the_look_at_point():
    from nerf_amd.utils.turntable import estimate_pivot
    centre = [0.12, -0.05, 0.2]
    poses = _ring_poses(17, centre, 0.8, 0.3)
    # every optical axis passes through `centre`: the least-squares point is that point whichever sign convention the
    # third column has (the distance to a line does not depend on the line's direction)
    assert float((estimate_pivot(poses) - torch.tensor(centre)).abs().max()) < 1e-5
    # a perturbed set against numpy's own least squares on the same normal equations
    g = torch.Generator().manual_seed(3)
    noisy = poses.clone()
    noisy[:, :3, 3] += 0.05 * torch.randn(17, 3, generator=g)
    C, D = noisy[:, :3, 3].double().numpy(), -noisy[:, :3, 2].double().numpy()
    A = np.eye(3)[None] - D[:, :, None] * D[:, None, :]
    ref = np.linalg.lstsq(A.sum(0), np.einsum("nij,nj->ni", A, C).sum(0), rcond=None)[0]
    assert np.abs(estimate_pivot(noisy).numpy() - ref).max() < 1e-5


def test_turntable_pose_rotates_position_and_frame_about_y():
    from nerf_amd.utils.turntable import turntable_pose
    start = _ring_poses(5, [0.0, 0.0, 0.0], 0.9, 0.25)[1]
    pivot = torch.tensor([0.1, 0.05, -0.2])
    assert float((turntable_pose(start, pivot, 0.0) - start).abs().max()) < 1e-7
    for t in (0.004, 0.25, 0.5, 0.9):
        th = 2 * math.pi * t
        Ry = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
        s = start.double().numpy()
        want = np.concatenate([Ry @ s[:, :3], (pivot.double().numpy() + Ry @ (s[:, 3] - pivot.double().numpy()))[:, None]], 1)
        got = turntable_pose(start, pivot, t)
        assert got.shape == (3, 4)
        assert np.abs(got.double().numpy() - want).max() < 1e-6
        assert abs(float((got[:, 3] - pivot).norm()) - float((start[:, 3] - pivot).norm())) < 1e-6


@pytest.mark.gpu
def test_turntable_frames_equal_the_oracle_render_of_the_same_poses(dev):
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.utils.turntable import estimate_pivot, render_turntable, turntable_pose
    torch.manual_seed(0)
    ref = O.FastNGPNeRFOracle(log2_hashmap_size=14)
    model = FastNGPNeRF(log2_hashmap_size=14)
    model.load_state_dict(ref.state_dict())
    model = model.to(dev)
    poses = _ring_poses(6, [0.0, 0.0, 0.0], 0.7, 0.2)
    pivot = estimate_pivot(poses)
    H, W, N, G, F = 12, 16, 24, 32, 4
    grid = (torch.rand(G ** 3, generator=torch.Generator().manual_seed(5)) > 0.3).to(torch.uint8)
    frames, cams = render_turntable(model, poses[2].to(dev), pivot.to(dev), H, W, (20.0, 21.0), pp=(7.5, 6.5), num_frames=F,
                                    near=0.05, far=1.2, n_samples=N, grid_res=G, density_grid=grid.to(dev))
    assert frames.shape == (F, H, W, 3) and frames.dtype == torch.uint8 and cams.shape == (F, 3, 4)
    from nerf_amd.render import xyz_to_idx
    for k in range(F):
        c2w = turntable_pose(poses[2], pivot, k / F)
        assert float((cams[k].cpu() - c2w).abs().max()) < 1e-6
        o, d = O.get_rays(c2w, (20.0, 21.0), H, W, pp=(7.5, 6.5))
        with torch.no_grad():
            pts, z = O.sample_points_along_rays(o, d, 0.05, 1.2, N, None)
            vd = (d / d.norm(dim=-1, keepdim=True))[..., None, :].expand_as(pts)
            rgb, sigma = ref(pts.reshape(-1, 3), vd.reshape(-1, 3))
            x, y, zz = ((pts.reshape(-1, 3) + 1) / 2 / (1.0 / G)).long().T.unbind(0)
            keep = grid[xyz_to_idx(x, y, zz).clamp(0, G ** 3 - 1)] == 1
            sigma = torch.where(keep, sigma.reshape(-1), torch.zeros(()))
            c_ref, _ = O.volume_render(rgb.view(H, W, N, 3), sigma.view(H, W, N), z.expand(H, W, N))
        want = (c_ref.clamp(0, 1) * 255)
        # bytes are truncated: a colour within 2e-5 of the oracle's can fall on the other side of an integer
        diff = (frames[k].cpu().float() - want.floor()).abs()
        assert float(diff.max()) <= 1.0 and float((diff > 0).float().mean()) < 0.01


@pytest.mark.gpu
def test_render_model_image_is_the_clamped_render(dev):
    from nerf_amd.Camera import Camera
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import render_image
    from nerf_amd.utils.turntable import render_model_image
    torch.manual_seed(1)
    model = NeRF().to(dev)
    cam = Camera(eye=torch.tensor([3.0, 1.0, 1.5]), target=torch.zeros(3), focal=(30.0, 30.0), H=10, W=12,
                 up=torch.tensor([0.0, 0.0, 1.0]))
    img = render_model_image(model, cam, 16)          # a camera on the host, as the reference's viewers build it
    o, d = Camera(eye=None, target=None, focal=(30.0, 30.0), H=10, W=12, c2w=cam.c2w.to(dev)).get_rays()
    want = (render_image(model, o, d, 2.0, 6.0, n_samples=16).clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
    assert img.dtype == np.uint8 and img.shape == (10, 12, 3) and np.array_equal(img, want)
