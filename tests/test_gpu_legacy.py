"""nerf_amd.utils.get_rays (drop-in for the older free functions of src/utils/get_rays.py) against golden
vectors produced by the reference's own functions."""
import pytest
import torch

from conftest import T, load_golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def test_legacy_helpers_match_reference(dev):
    from nerf_amd.utils import get_rays as L
    g = load_golden("g8_legacy.npz")
    eye, tgt = T(g["eye"]).to(dev), T(g["tgt"]).to(dev)
    up = L.compute_stable_up(eye, tgt)
    torch.testing.assert_close(up.cpu(), T(g["up"]), rtol=0, atol=1e-6)
    up_v = L.compute_stable_up(torch.tensor([0.0, -3.0, 0.0], device=dev), torch.zeros(3, device=dev))
    torch.testing.assert_close(up_v.cpu(), T(g["up_vert"]), rtol=0, atol=1e-6)
    c2w = L.look_at_camera(eye, tgt, up)
    torch.testing.assert_close(c2w.cpu(), T(g["c2w"]), rtol=0, atol=1e-6)
    H, W, focal = int(g["hwf"][0]), int(g["hwf"][1]), float(g["hwf"][2])
    o, d = L.get_rays(H, W, focal, T(g["c2w"]).to(dev))
    assert o.shape == (W, H, 3) and d.shape == (W, H, 3)                       # width-major, as the reference
    assert torch.equal(o.cpu(), T(g["rays_o"]))
    torch.testing.assert_close(d.cpu(), T(g["rays_d"]), rtol=0, atol=3e-7)
    near, far, n = float(g["cfg"][0]), float(g["cfg"][1]), int(g["cfg"][2])
    pts, z = L.sample_points_along_rays(T(g["rays_o"]).to(dev), T(g["rays_d"]).to(dev), near, far, n)
    assert torch.equal(z.cpu(), T(g["z"])) and torch.equal(pts.cpu(), T(g["pts"]))   # same inputs: bit-exact
    for name in ("complex", "sphere", "torus"):
        rgb, sigma = getattr(L, "evaluate_fake_radiance_field_" + name)(T(g["pts"]).to(dev))
        # the colour switches where two densities cross: the GPU's exp may put a point on the other side
        assert float((rgb.cpu() != T(g[name + "_rgb"])).any(dim=-1).float().mean()) < 2e-3, name
        # exp / sqrt of the GPU differ from the CPU's in the last bit
        torch.testing.assert_close(sigma.cpu(), T(g[name + "_sigma"]), rtol=1e-5, atol=1e-7)
        img = L.volume_render(T(g[name + "_rgb"]).to(dev), T(g[name + "_sigma"]).to(dev), T(g["z"]).to(dev))
        assert img.shape == (W, H, 3)
        torch.testing.assert_close(img.cpu(), T(g[name + "_img"]), rtol=0, atol=2e-6)


def test_legacy_volume_render_tail_differs_from_camera(dev):
    """1e10 vs 1e5 after the last sample (src/utils/get_rays.py:214 vs src/Camera.py:154): visible where the
    last density is tiny."""
    from nerf_amd.Camera import Camera
    from nerf_amd.utils import get_rays as L
    g6 = load_golden("g6_composite.npz")
    rgb, sigma, z = (T(g6[k]).to(dev) for k in ("a_rgb", "a_sigma", "a_z"))
    img = L.volume_render(rgb, sigma, z)
    torch.testing.assert_close(img.cpu(), T(g6["legacy_c"]), rtol=0, atol=2e-6)
    sig = torch.zeros_like(sigma)
    sig[..., -1] = 1e-8                                     # alpha_last = 1 - exp(-1e-8 * tail)
    a = L.volume_render(rgb, sig, z)
    b, _ = Camera.volume_render(rgb, sig, z)
    assert float((a - b).abs().max()) > 0.1


def test_preload_llff_writes_the_three_splits(dev, tmp_path):
    """src/data/preload_llff_data.py: poses_bounds.npy + images/ -> train/val/test .pt files with the older
    helpers' rays ([W,H,3], +z forward), the images and per-pixel bounds.  Ten frames: 8 / 1 / 1."""
    import numpy as np
    from PIL import Image
    from nerf_amd.data.preload_llff_data import load_llff_data, precompute_and_split_llff, split_dataset_indices
    rng = np.random.default_rng(0)
    N, H, W = 10, 6, 8
    table = np.zeros((N, 17))
    for i in range(N):
        ang = 0.2 * i
        pose = np.array([[np.cos(ang), 0, np.sin(ang), 0.1 * i], [0, 1, 0, 0.2], [-np.sin(ang), 0, np.cos(ang), 1.0]])
        table[i, :15] = np.concatenate([pose, np.array([[H], [W], [11.0 + i]])], axis=1).reshape(-1)
        table[i, 15:] = [1.0 + i, 9.0 + i]
    (tmp_path / "images").mkdir()
    np.save(tmp_path / "poses_bounds.npy", table)
    imgs = rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)
    for i in range(N):
        Image.fromarray(imgs[i]).save(tmp_path / "images" / ("%03d.png" % i))
    poses, bounds, images = load_llff_data(str(tmp_path))
    assert poses.shape == (N, 3, 5) and bounds.shape == (N, 2) and len(images) == N
    assert torch.equal(images[3], torch.from_numpy(imgs[3].astype(np.float32) / 255.0))
    np.random.seed(5)
    want_split = split_dataset_indices(N)
    assert [len(s) for s in want_split] == [8, 1, 1] and sorted(sum(want_split, [])) == list(range(N))
    np.random.seed(5)
    precompute_and_split_llff(str(tmp_path), str(tmp_path / "parts"), H=H, W=W, device=dev)
    for name, idx in zip(("train", "val", "test"), want_split):
        part = torch.load(tmp_path / "parts" / (name + ".pt"), weights_only=False)
        assert part["rays_o"].shape == (len(idx), W, H, 3) and part["near"].shape == (len(idx), H, W)
        assert float(part["focal"]) == 11.0
        for k, i in enumerate(idx):
            c2w = torch.eye(4)
            c2w[:3, :4] = torch.from_numpy(poses[i][:, :4])
            o, d = O.legacy_get_rays(H, W, float(poses[i, -1, -1]), c2w)
            assert torch.equal(part["rays_o"][k], o.contiguous())
            assert float((part["rays_d"][k] - d).abs().max()) < 1e-6
            assert torch.equal(part["rgb"][k], images[i])
            assert float(part["near"][k][0, 0]) == bounds[i, 0] and float(part["far"][k][-1, -1]) == bounds[i, 1]
