"""The multi-camera ray batch against the single-camera raygen (bit-exact) and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n=7, H=24, W=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    poses = torch.zeros(n, 3, 4)
    for i in range(n):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        poses[i, :, :3] = q
        poses[i, :, 3] = torch.randn(3, generator=g)
    intr = torch.stack([torch.tensor([30.0 + i, 31.0 + 0.5 * i, W / 2 + 0.25 * i, H / 2 - 0.125 * i]) for i in range(n)])
    images = torch.rand(n, H, W, 3, generator=g)
    return poses, intr, images


def test_raygen_multi_matches_per_camera_raygen_and_oracle(dev):
    from nerf_amd import ops
    from nerf_amd.data.ray_buffer import camera_rows
    from oracle import nerf_oracle as O
    poses, intr, images = _scene()
    n, H, W = images.shape[:3]
    g = torch.Generator().manual_seed(1)
    R = 5000
    cam_id = torch.randint(0, n, (R,), generator=g, dtype=torch.int32)
    pix = torch.randint(0, H * W, (R,), generator=g, dtype=torch.int32)
    o, d, rgb = ops.raygen_multi(camera_rows(poses, intr).to(dev), H, W, cam_id.to(dev), pix.to(dev), images.to(dev))
    torch.testing.assert_close(rgb.cpu(), images.view(n, H * W, 3)[cam_id.long(), pix.long()], rtol=0, atol=0)
    for c in range(n):
        sel = (cam_id == c).nonzero().squeeze(1)
        fx, fy, cx, cy = intr[c].tolist()
        o1, d1 = ops.raygen(poses[c].to(dev), fx, fy, cx, cy, H, W, pix[sel].to(dev))
        assert torch.equal(o[sel.to(dev)], o1) and torch.equal(d[sel.to(dev)], d1)
        oo, od = O.get_rays(poses[c], (fx, fy), H, W, pp=(cx, cy))
        torch.testing.assert_close(d[sel.to(dev)].cpu(), od.reshape(-1, 3)[pix[sel].long()], rtol=0, atol=2e-6)
        torch.testing.assert_close(o[sel.to(dev)].cpu(), oo.reshape(-1, 3)[pix[sel].long()], rtol=0, atol=0)
    # without images: rays only; empty batch
    o2, d2 = ops.raygen_multi(camera_rows(poses, intr).to(dev), H, W, cam_id.to(dev), pix.to(dev))
    assert torch.equal(o2, o) and torch.equal(d2, d)
    e = ops.raygen_multi(camera_rows(poses, intr).to(dev), H, W, cam_id[:0].to(dev), pix[:0].to(dev))
    assert e[0].shape == (0, 3)


def test_ray_buffer_epoch_covers_every_pixel_once_across_ranks(dev):
    from nerf_amd.data import RayBuffer
    poses, intr, images = _scene(n=3, H=8, W=16)
    total = 3 * 8 * 16
    seen = []
    for rank in range(2):
        rb = RayBuffer(images, poses, intr, dev, rank=rank, world=2, seed=5)
        for _ in range(total // (2 * 64)):
            o, d, rgb = rb.next_batch(64)
            assert o.shape == (64, 3) and d.shape == (64, 3) and rgb.shape == (64, 3)
            seen.append(rgb.cpu())
    seen = torch.cat(seen)
    # every pixel of every image exactly once: compare as sorted rows
    a = seen[np.lexsort(seen.numpy().T[::-1])]
    allpix = images.view(-1, 3)
    b = allpix[np.lexsort(allpix.numpy().T[::-1])]
    assert torch.equal(a, b)


def test_out_of_range_ids_give_nan_rays_not_faults(dev):
    """A camera or pixel id outside its range reads nothing and comes back as NaN."""
    from nerf_amd import ops
    cams = torch.zeros(2, 16)
    cams[:, 0] = cams[:, 5] = cams[:, 10] = 1.0
    cams[:, 12:] = torch.tensor([50.0, 50.0, 4.0, 3.0])
    H, W = 6, 8
    images = torch.rand(2, H, W, 3)
    cam_id = torch.tensor([0, 1, 2, -1, 1, 0], dtype=torch.int32)
    pix = torch.tensor([0, 47, 5, 5, 48, -3], dtype=torch.int32)
    o, d, rgb = ops.raygen_multi(cams.to(dev), H, W, cam_id.to(dev), pix.to(dev), images.to(dev))
    bad = torch.tensor([False, False, True, True, True, True])
    assert torch.isnan(d.cpu()[bad]).all() and torch.isnan(rgb.cpu()[bad]).all() and torch.isnan(o.cpu()[bad]).all()
    assert torch.isfinite(d.cpu()[~bad]).all() and torch.equal(rgb.cpu()[1], images[1].reshape(-1, 3)[47])
    o1, d1 = ops.raygen(torch.eye(4)[:3].to(dev), 50.0, 50.0, 4.0, 3.0, H, W, torch.tensor([0, 48, -1], dtype=torch.int32).to(dev))
    assert torch.isfinite(d1.cpu()[0]).all() and torch.isnan(d1.cpu()[1:]).all()
