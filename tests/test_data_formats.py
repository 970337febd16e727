"""Dataset formats: transforms.json / poses_bounds.npy parsing, unit-sphere normalisation and the
intrinsics rescaling, checked against independent numpy restatements of the reference's formulas
(src/data/nerf_capture_dataset.py:20-111, src/data/LLFFDataset.py:24-86).  The reference's own classes need
torchvision and imageio, which are not installed here, so these results are not pinned by running them:
parity unpinned beyond the formulas."""
import json
import os

import numpy as np
import pytest
import torch

from nerf_amd.data import LLFFDataset, NeRFCaptureDataset, normalize_poses, read_poses_bounds

PIL = pytest.importorskip("PIL.Image")


def _write_capture_scene(root, n=12, w=64, h=48, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    frames = []
    for i in range(n):
        m = np.eye(4)
        m[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        m[:3, 3] = rng.normal(size=3) * 3 + np.array([1.0, -2.0, 0.5])
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        PIL.fromarray(img).save(os.path.join(root, "images", f"{i}.png"))
        frames.append({"file_path": f"images/{i}", "cx": w / 2 + 0.25 * i, "cy": h / 2 - 0.5, "fl_x": 50.0 + i, "fl_y": 51.0 + i,
                       "w": w, "h": h, "transform_matrix": m.tolist()})
    with open(os.path.join(root, "transforms.json"), "w") as f:
        json.dump({"frames": frames}, f)
    return frames


def test_capture_normalisation_and_intrinsics(tmp_path):
    frames = _write_capture_scene(str(tmp_path))
    torch.manual_seed(3)
    ds = NeRFCaptureDataset(str(tmp_path), split="train", img_wh=(32, 24))
    # unit-sphere normalisation, restated in float64 numpy
    c = np.array([np.array(fr["transform_matrix"])[:3, 3] for fr in frames])
    centre = c.mean(0)
    scale = 1.2 * np.linalg.norm(c - centre, axis=1).max()
    want = (c - centre) / scale
    got = torch.stack([p[:3, 3] for p in ds.poses]).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    np.testing.assert_allclose(ds.scene_center.numpy(), centre, atol=2e-6)
    assert abs(float(ds.scene_scale) - scale) < 1e-5
    assert np.linalg.norm(got, axis=1).max() <= 1 / 1.2 + 1e-6
    # rotations untouched
    for p, fr in zip(ds.poses, frames):
        np.testing.assert_allclose(p[:3, :3].numpy(), np.array(fr["transform_matrix"])[:3, :3], atol=1e-6)
    # split: 10 % validation, disjoint, complete
    assert len(ds.val_idx) == 1 and len(ds.train_idx) == 11
    assert sorted(ds.val_idx + ds.train_idx) == list(range(12))
    assert len(ds) == 11
    img, pose, focal, pp, wh = ds[0]
    k = ds.train_idx[0]
    assert img.shape == (3, 24, 32) and img.dtype == torch.float32 and 0 <= float(img.min()) and float(img.max()) <= 1
    assert pose.shape == (4, 4) and pose is ds.poses[k]
    np.testing.assert_allclose(focal.numpy(), [frames[k]["fl_x"] * 0.5, frames[k]["fl_y"] * 0.5], rtol=1e-6)
    np.testing.assert_allclose(pp.numpy(), [frames[k]["cx"] * 0.5, frames[k]["cy"] * 0.5], rtol=1e-6)
    assert wh.tolist() == [32.0, 24.0]
    assert len(NeRFCaptureDataset(str(tmp_path), split="val", img_wh=(32, 24))) == 1
    assert NeRFCaptureDataset(str(tmp_path), subset=5).N_images == 5


def test_capture_image_is_pil_bilinear(tmp_path):
    _write_capture_scene(str(tmp_path), n=10)
    ds = NeRFCaptureDataset(str(tmp_path), img_wh=(64, 48))    # same size: the pixels come through unchanged
    img = ds[0][0]
    src = np.asarray(PIL.open(ds.image_paths[ds.train_idx[0]]).convert("RGB"))
    np.testing.assert_array_equal((img.permute(1, 2, 0) * 255).round().numpy().astype(np.uint8), src)


def test_normalize_poses_single_camera_is_nan_like_reference():
    # one camera: the largest distance is 0 and the reference divides by it
    p = [torch.eye(4)]
    normalize_poses(p)
    assert torch.isnan(p[0][:3, 3]).all()


def _write_llff_scene(root, n=5, h=20, w=30, factor=8, seed=1):
    rng = np.random.default_rng(seed)
    pb = rng.normal(size=(n, 17))
    pb[:, 4] = h
    pb[:, 9] = w
    pb[:, 14] = 123.5
    pb[:, 15:] = np.abs(pb[:, 15:]) + [[1.0, 5.0]]
    np.save(os.path.join(root, "poses_bounds.npy"), pb)
    d = os.path.join(root, f"images_{factor}")
    os.makedirs(d, exist_ok=True)
    imgs = []
    for i in range(n):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        PIL.fromarray(img).save(os.path.join(d, f"im{i:02d}.png"))
        imgs.append(img)
    return pb, np.stack(imgs)


def test_llff_small_images(tmp_path):
    pb, imgs = _write_llff_scene(str(tmp_path))
    poses, bounds, focal = read_poses_bounds(tmp_path / "poses_bounds.npy")
    assert poses.shape == (5, 3, 5) and bounds.shape == (5, 2) and focal == 123.5
    np.testing.assert_array_equal(poses.reshape(5, 15), pb[:, :15])
    ds = LLFFDataset(str(tmp_path), split="train")
    assert (ds.H, ds.W, ds.near, ds.far, ds.focal) == (20, 30, 2.0, 6.0, 123.5) and len(ds) == 5
    image, pose, f, H, W, b = ds[2]
    np.testing.assert_allclose(image.numpy(), imgs[2] / 255.0, atol=1e-7)     # narrower than 100: not resized
    np.testing.assert_array_equal(pose.numpy(), pb[2, :15].reshape(3, 5)[:, :4].astype(np.float32))
    np.testing.assert_array_equal(b, pb[2, 15:])
    assert len(LLFFDataset(str(tmp_path), split="val")) == 0
    with pytest.raises(ValueError):
        LLFFDataset(str(tmp_path), split="test")
    with pytest.raises(FileNotFoundError):
        LLFFDataset(str(tmp_path), factor=4)


def test_llff_wide_images_are_resized_and_transposed(tmp_path):
    _write_llff_scene(str(tmp_path), n=2, h=60, w=200)
    ds = LLFFDataset(str(tmp_path))
    assert (ds.H, ds.W) == (30, 100) and ds.focal == 123.5      # the focal length is not rescaled (as in the reference)
    assert tuple(ds[0][0].shape) == (100, 30, 3)                # [W, H, 3]: the reference's permute(2, 1, 0)


def test_ssim_follows_the_published_box_window_definition():
    """scikit-image is not installed here, so the reference's own call (src/train_nerf.py:174) cannot be run:
    parity unpinned.  Checked against the algorithm scikit-image documents -- uniform_filter means, sample covariances,
    the border of (win_size - 1) / 2 pixels cropped before averaging -- restated with scipy."""
    import numpy as np
    import scipy.ndimage as ndi
    import torch
    from nerf_amd.render import ssim
    rng = np.random.default_rng(0)
    gt = rng.random((37, 50, 3)).astype(np.float32)
    img = np.clip(gt + 0.1 * rng.standard_normal(gt.shape).astype(np.float32), 0, 1)

    def restated(a, b, win=7, R=1.0):
        vals = []
        for c in range(a.shape[2]):
            x, y = a[..., c].astype(np.float64), b[..., c].astype(np.float64)
            f = lambda t: ndi.uniform_filter(t, size=win)
            ux, uy = f(x), f(y)
            k = win * win / (win * win - 1.0)
            vx, vy, vxy = k * (f(x * x) - ux * ux), k * (f(y * y) - uy * uy), k * (f(x * y) - ux * uy)
            c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
            s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
            p = (win - 1) // 2
            vals.append(s[p:-p, p:-p].mean())
        return float(np.mean(vals))

    got = float(ssim(torch.from_numpy(gt), torch.from_numpy(img)))
    assert abs(got - restated(gt, img)) < 1e-9
    assert abs(float(ssim(torch.from_numpy(gt), torch.from_numpy(gt))) - 1.0) < 1e-12
    assert 0.0 < got < 1.0
