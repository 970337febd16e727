"""Pin the CPU oracle to golden vectors produced by the reference itself
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import T, load_golden, rel_err
from oracle import nerf_oracle as O


def test_raygen_matches_reference():
    g = load_golden("g1_raygen.npz")
    H, W = [int(v) for v in g["robot_hw"]]
    fx, fy, cx, cy = [float(v) for v in g["robot_intr"]]
    o, d = O.get_rays(T(g["robot_c2w"]), (fx, fy), H, W, pp=(cx, cy))
    assert torch.equal(o, T(g["robot_o"])) and torch.equal(d, T(g["robot_d"]))
    c2w = O.look_at_camera(torch.tensor([1.0, 1.0, 2.0]), torch.zeros(3))
    assert torch.equal(c2w, T(g["look_c2w"]))
    o, d = O.get_rays(c2w, (100.0, 100.0), 100, 100)
    assert torch.equal(o, T(g["look_o"])) and torch.equal(d, T(g["look_d"]))
    # the reference's look-at camera shoots away from its target (rays follow -forward)
    assert float(d[50, 50] @ torch.tensor([1.0, 1.0, 2.0])) > 0
    assert torch.equal(O.look_at_camera(torch.tensor([0.0, 3.0, 0.0]), torch.zeros(3)), T(g["look_up_z"]))
    H, W = [int(v) for v in g["llff_hw"]]
    f = float(g["llff_f"][0])
    o, d = O.get_rays(T(g["llff_c2w"]), (f, f), H, W)
    assert torch.equal(d, T(g["llff_d"]))


def test_stratified_bit_exact():
    g = load_golden("g2_stratified.npz")
    ro, rd = T(g["rays_o"]), T(g["rays_d"])
    for tag in "abc":
        near, far, n = g[f"{tag}_cfg"]
        pts, z = O.sample_points_along_rays(ro, rd, float(near), float(far), int(n), T(g[f"{tag}_u"]))
        assert torch.equal(z, T(g[f"{tag}_z"])), tag
        assert torch.equal(pts, T(g[f"{tag}_pts"])), tag
        pts0, z0 = O.sample_points_along_rays(ro, rd, float(near), float(far), int(n), None)
        assert torch.equal(z0.contiguous(), T(g[f"{tag}_z0"])) and torch.equal(pts0, T(g[f"{tag}_pts0"]))
        # the sort after jittering never reorders: strata are disjoint
        assert bool((z[..., 1:] >= z[..., :-1]).all())


def test_positional_encoding():
    g = load_golden("g3_pe.npz")
    x = T(g["x"])
    assert torch.equal(O.positional_encoding(x, 10), T(g["pe10"]))
    assert torch.equal(O.positional_encoding(x, 4), T(g["pe4"]))


def _filled_tables(sizes):
    tabs = []
    for l, n in enumerate(sizes):
        r = torch.arange(int(n), dtype=torch.float64)
        tabs.append(torch.stack([(torch.sin(r * (0.37 + 0.011 * l) + 1.3 * f + 0.5 * l) * 0.5).float()
                                 for f in range(2)], dim=1))
    return tabs


def test_hash_grid_indices_and_features():
    g = load_golden("g4_hash.npz")
    res = O.hash_resolutions()
    assert res == [int(v) for v in g["resolutions"]] == [16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256,
                                                          322, 406, 512]
    sizes = O.hash_table_sizes(res)
    assert sizes == [int(v) for v in g["sizes"]]
    x = T(g["x"])
    for l, (r, s) in enumerate(zip(res, sizes)):
        idx, w = O.hash_corner_indices(x, r, s)
        assert torch.equal(idx.t().to(torch.int32), T(g["idx"][l])), l
        assert torch.equal(w, T(g["w"][l])), l
    tabs = [t.requires_grad_(True) for t in _filled_tables(sizes)]
    feats = O.hash_encode(x, tabs, res)
    assert torch.equal(feats.detach(), T(g["feats"]))
    (feats * T(g["a"])).sum().backward()
    for l, t in enumerate(tabs):
        assert abs(float(t.grad.double().norm()) - g["grad_l2"][l]) <= 1e-6 * g["grad_l2"][l]
        rows = T(g["grad_rows"][l]).long()
        assert torch.allclose(t.grad[rows], T(g["grad_vals"][l]), rtol=1e-5, atol=1e-7)


def _check_model(g, name, model, x, d):
    for k, v in model.state_dict().items():
        ck = g[f"{name}_ck/{k}"] if f"{name}_ck/{k}" in g else None
        if ck is not None:
            assert float(v.double().sum()) == ck[0] and float(v.double().abs().sum()) == ck[1], k


def _check_outputs(g, name, model, x, d, tol=1e-6):
    rgb, sigma = model(x, d)
    assert rgb.shape == g[f"{name}_rgb"].shape and sigma.shape == g[f"{name}_sigma"].shape
    assert torch.allclose(rgb, T(g[f"{name}_rgb"]), rtol=0, atol=tol)
    assert torch.allclose(sigma, T(g[f"{name}_sigma"]), rtol=1e-6, atol=tol)
    loss = (rgb * T(g[f"{name}_a"])).sum() + (sigma * T(g[f"{name}_b"])).sum()
    model.zero_grad()
    loss.backward()
    for k, p in model.named_parameters():
        gr = p.grad.reshape(-1)
        ref_n = float(g[f"{name}_gnorm/{k}"][0])
        assert abs(float(gr.double().norm()) - ref_n) <= 1e-5 * ref_n + 1e-12, k
        ii = T(g[f"{name}_gidx/{k}"]).long()
        assert torch.allclose(gr[ii], T(g[f"{name}_gval/{k}"]), rtol=1e-4, atol=1e-5 * ref_n + 1e-12), k


def test_vanilla_models_seeded_init_and_grads():
    g = load_golden("g5_mlp.npz")
    x, d = T(g["x"]), T(g["d"])
    torch.manual_seed(0)
    m = O.NeRFOracle()
    _check_model(g, "nerf", m, x, d)
    _check_outputs(g, "nerf", m, x, d)
    torch.manual_seed(0)
    m = O.FastMLPNeRFOracle()
    _check_model(g, "fastmlp", m, x, d)
    _check_outputs(g, "fastmlp", m, x, d)


def test_ngp_model_seeded_init_and_grads():
    g = load_golden("g5_ngp.npz")
    x, d = T(g["x"]), T(g["d"])
    torch.manual_seed(0)
    m = O.FastNGPNeRFOracle()
    _check_model(g, "ngp", m, x, d)
    assert sum(p.numel() for p in m.parameters()) == 10442933
    _check_outputs(g, "ngp", m, x, d)
    assert torch.allclose(m.query_density(x), T(g["ngp_density"]), rtol=1e-6, atol=0)
    rgb0, _ = m(x[:16], None)
    assert torch.allclose(rgb0, T(g["ngp_nodir_rgb"]), atol=1e-7)
    with torch.no_grad():
        for k, p in m.named_parameters():
            for pat, s in zip(g["boost_keys"], g["boost_vals"]):
                if k.startswith(str(pat)):
                    p.mul_(float(s))
    _check_outputs(g, "ngpb", m, x, d)


def test_volume_render_and_grads():
    g = load_golden("g6_composite.npz")
    for tag in "ac":
        for wb in (0, 1):
            r = T(g[f"{tag}_rgb"]).requires_grad_(True)
            s = T(g[f"{tag}_sigma"]).requires_grad_(True)
            c, w = O.volume_render(r, s, T(g[f"{tag}_z"]), white_bkgd=bool(wb))
            k = f"{tag}{wb}"
            assert torch.equal(c.detach(), T(g[f"{k}_c"])) and torch.equal(w.detach(), T(g[f"{k}_w"]))
            loss = torch.nn.functional.mse_loss(c, T(g[f"{tag}_tgt"])) + 0.05 * (1 - w.sum(dim=-1)).abs().mean()
            loss.backward()
            assert torch.allclose(r.grad, T(g[f"{k}_drgb"]), rtol=1e-6, atol=1e-9)
            assert torch.allclose(s.grad, T(g[f"{k}_dsigma"]), rtol=1e-6, atol=1e-9)


def test_analytic_scene():
    g = load_golden("g7_scene.npz")
    H, W, f, near, far, n = g["cfg"]
    H, W, n = int(H), int(W), int(n)
    for v in range(g["c2w"].shape[0]):
        o, d = O.get_rays(T(g["c2w"][v]), (float(f), float(f)), H, W)
        pts, z = O.sample_points_along_rays(o, d, float(near), float(far), n, None)
        rgb, sigma = O.analytic_field(pts)
        img, _ = O.volume_render(rgb, sigma * 5.0, z)
        assert torch.allclose(img, T(g["img"][v]), atol=1e-6)
    assert float(T(g["img"]).max()) > 0.2  # the object is in view


def test_sample_pdf_properties():
    """Hierarchical sampling is build-defined (parity unpinned): check the
    restatement against an independent float64 formulation."""
    gen = torch.Generator().manual_seed(11)
    R, N, Nf = 64, 64, 128
    z = torch.sort(torch.rand(R, N, generator=gen) * 4 + 2, dim=-1).values
    w = torch.rand(R, N, generator=gen) ** 4
    w[0] = 0.0           # flat pdf
    w[1, 1:-1] = 0.0
    w[1, 20] = 1.0       # one spike
    u = torch.rand(R, Nf, generator=gen)
    u[2, 0] = 0.0
    zf, idx, z_all = O.sample_pdf(z, w, u)
    assert zf.shape == (R, Nf) and idx.dtype == torch.int32 and z_all.shape == (R, N + Nf)
    assert bool((z_all[:, 1:] >= z_all[:, :-1]).all())
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    assert bool((zf >= bins[:, :1] - 1e-6).all()) and bool((zf <= bins[:, -1:] + 1e-6).all())
    # float64 reference using cumsum
    wd = (w[:, 1:-1].double() + 1e-5)
    cdf = torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(wd / wd.sum(-1, keepdim=True), -1)], -1)
    idx64 = torch.searchsorted(cdf, u.double().contiguous(), right=True)
    agree = (idx64 == idx.long()).float().mean()
    assert agree > 0.999
    # spike ray: nearly every fine sample lands in the spike's bin
    lo, hi = bins[1, 19], bins[1, 20]
    inside = ((zf[1] >= lo) & (zf[1] <= hi)).float().mean()
    assert inside > 0.95
    assert rel_err(O.wave_inclusive_scan(w), torch.cumsum(w.double(), -1)) < 1e-6


def test_ndc_rays_maps_near_plane_to_minus_one():
    o = torch.tensor([[0.1, -0.2, 0.3], [0.0, 0.0, 0.0]])
    d = torch.tensor([[0.1, 0.2, -1.0], [0.0, 0.0, -1.0]])
    o2, d2 = O.ndc_rays(378, 504, 415.0, 1.0, o, d)
    assert torch.allclose(o2[:, 2], torch.tensor([-1.0, -1.0]), atol=1e-6)
    assert torch.allclose((o2 + d2)[:, 2], torch.tensor([1.0, 1.0]), atol=1e-6)  # t=1 is infinity


def test_legacy_free_functions_match_reference():
    """The older helpers of src/utils/get_rays.py: +z camera, [W,H,3] rays, 1e10 tail."""
    g = load_golden("g8_legacy.npz")
    eye, tgt = T(g["eye"]), T(g["tgt"])
    up = O.legacy_compute_stable_up(eye, tgt)
    assert torch.equal(up, T(g["up"]))
    assert torch.equal(O.legacy_compute_stable_up(torch.tensor([0.0, -3.0, 0.0]), torch.zeros(3)), T(g["up_vert"]))
    c2w = O.legacy_look_at_camera(eye, tgt, up)
    assert torch.equal(c2w, T(g["c2w"]))
    H, W, focal = int(g["hwf"][0]), int(g["hwf"][1]), float(g["hwf"][2])
    o, d = O.legacy_get_rays(H, W, focal, c2w)
    assert o.shape == (W, H, 3) and torch.equal(o, T(g["rays_o"])) and torch.equal(d, T(g["rays_d"]))
    near, far, n = float(g["cfg"][0]), float(g["cfg"][1]), int(g["cfg"][2])
    pts, z = O.sample_points_along_rays(o, d, near, far, n, None)
    assert torch.equal(pts, T(g["pts"])) and torch.equal(z.contiguous(), T(g["z"]))
    for name, fn in (("complex", O.analytic_field), ("sphere", O.analytic_field_sphere), ("torus", O.analytic_field_torus)):
        rgb, sigma = fn(pts)
        assert torch.equal(rgb.contiguous(), T(g[name + "_rgb"])) and torch.equal(sigma, T(g[name + "_sigma"])), name
        assert torch.equal(O.legacy_volume_render(rgb, sigma, z), T(g[name + "_img"])), name
    g6 = load_golden("g6_composite.npz")
    assert torch.equal(O.legacy_volume_render(T(g6["a_rgb"]), T(g6["a_sigma"]), T(g6["a_z"])), T(g6["legacy_c"]))


def test_forced_relu_masks_reproduce_the_network():
    """Forcing the ReLU pattern the network takes by itself changes nothing; the recorded pre-activations
    are what the pattern is derived from."""
    torch.manual_seed(3)
    m = O.NeRFOracle()
    x, d = torch.randn(40, 3), torch.nn.functional.normalize(torch.randn(40, 3), dim=-1)
    rec = []
    rgb, sigma = m(x, d, record=rec)
    assert [r.shape[1] for r in rec] == [256] * 8 + [128]
    masks = torch.cat([r > 0 for r in rec], dim=1)
    rgb_m, sigma_m = m(x, d, masks=masks)
    assert torch.equal(rgb, rgb_m) and torch.equal(sigma, sigma_m)
    a, b = m(x, d)
    assert torch.equal(rgb, a) and torch.equal(sigma, b)


def test_occupancy_mask_semantics():
    """sigma is zeroed where the grid cell is empty, cells are indexed with res = 32 whatever grid_res is."""
    grid = torch.zeros(32 ** 3, dtype=torch.uint8)
    pts = torch.tensor([[-1.0, -1.0, -1.0], [0.0, 0.0, 0.0], [0.99, 0.99, 0.99], [-0.5, 0.25, 0.75]])
    cells = ((pts + 1) / 2 * 32).long()
    grid[O.xyz_to_idx(cells[1, 0], cells[1, 1], cells[1, 2])] = 1
    grid[O.xyz_to_idx(cells[3, 0], cells[3, 1], cells[3, 2])] = 1
    sigma = torch.tensor([1.0, 2.0, 3.0, 4.0], requires_grad=True)
    out = O.occupancy_mask(sigma, pts, grid, 32)
    assert out.tolist() == [0.0, 2.0, 0.0, 4.0]
    out.sum().backward()
    assert sigma.grad.tolist() == [0.0, 1.0, 0.0, 1.0]


def test_g10_schedule_fixture_rates_scene_and_first_steps():
    """g10_sched.npz: the learning rates are those of ExponentialLR(0.98) / CosineAnnealingLR(30, 1e-4) stepped per
    epoch, the ground-truth images are the oracle's, and the oracle's own training step (its NeRF, sampling,
    compositing, torch Adam and clipping) reproduces the first steps of the vanilla trajectory."""
    import math
    g = load_golden("g10_sched.npz")
    a_lr = np.array([5e-4 * 0.98 ** (s // 8) for s in range(80)])
    b_lr = np.array([1e-4 + (1e-2 - 1e-4) * (1 + math.cos(math.pi * (s // 2) / 30)) / 2 for s in range(40)])
    assert np.allclose(g["a_lr"], a_lr, rtol=1e-9) and np.allclose(g["b_lr"], b_lr, rtol=1e-9)
    scenes = {}
    for tag in ("a", "b"):
        H, W, f, near, far, n, boost, views, shrink = [float(v) for v in g[tag + "_cfg"]]
        H, W, n = int(H), int(W), int(n)
        rays, imgs = [], []
        for k in range(int(views)):
            o, d = O.get_rays(T(g[tag + "_c2w"][k]), (f, f), H, W)
            pts, z = O.sample_points_along_rays(o, d, near, far, n, None)
            rgb, sigma = O.analytic_field(pts * shrink)
            img, _ = O.volume_render(rgb, sigma * boost, z)
            want = float(g[tag + "_img_sum"][k])
            assert abs(float(img.double().sum()) - want) < 1e-6 * max(1.0, want), (tag, k)
            rays.append((o.reshape(-1, 3), d.reshape(-1, 3)))
            imgs.append(img.reshape(-1, 3))
        scenes[tag] = (H, W, near, far, n, rays, imgs)
    H, W, near, far, n, rays, imgs = scenes["a"]
    torch.manual_seed(0)
    m = O.NeRFOracle()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    for step in range(2):
        torch.manual_seed(2000 + step)
        u = torch.rand(H, W, n).reshape(H * W, n)
        c_map, _, _, _ = O.render_rays(m, rays[step][0], rays[step][1], near, far, n, t_rand=u, chunk=65536)
        loss = torch.nn.functional.mse_loss(c_map, imgs[step])
        assert abs(float(loss) - float(g["a_loss"][step])) < 1e-5 * float(g["a_loss"][step]), step
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
        opt.step()
