"""Occupancy-driven sample skipping: the model runs on the samples in occupied cells only.  The reference zeroes the
densities of the other samples after the model has run on them (src/train_nerf_ngp.py:113-117,
src/utils/analyze_capture_results.py:283-287); a sample with zero density has zero weight and no gradient, so the
renders must be identical and the parameter gradients equal up to summation order."""
import pytest
import torch

from conftest import rel_err
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def _ball_grid(res, radius):
    """uint8 [res^3], indexed x + 32 y + 1024 z as the reference's xyz_to_idx does: cells whose centre lies in a ball."""
    c = (torch.arange(res).float() + 0.5) / res * 2 - 1
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    occ = ((x ** 2 + y ** 2 + z ** 2) < radius ** 2)
    grid = torch.zeros(res ** 3, dtype=torch.uint8)
    ix, iy, iz = torch.meshgrid(torch.arange(res), torch.arange(res), torch.arange(res), indexing="ij")
    grid[(ix + iy * 32 + iz * 1024).reshape(-1)] = occ.reshape(-1).to(torch.uint8)
    return grid


@pytest.mark.parametrize("P,dirs_div", [(1, 1), (1023, 1), (1024, 64), (70001, 7), (1 << 20, 64)])
def test_compaction_keeps_the_occupied_samples_in_order(dev, P, dirs_div):
    from nerf_amd import ops
    gen = torch.Generator().manual_seed(P)
    grid = _ball_grid(32, 0.7)
    pts = torch.rand(P, 3, generator=gen) * 2.4 - 1.2          # some outside the grid: clamped like the mask kernel
    dirs = torch.randn((P + dirs_div - 1) // dirs_div, 3, generator=gen)
    pd, gd = pts.to(dev), grid.to(dev)
    ones = torch.ones(P, device=dev)
    mask = ops.occupancy_mask(ones, pd, gd, 32) == 1.0          # the mask kernel's own decision per sample
    want = torch.nonzero(mask).reshape(-1).to(torch.int32)
    kept, slot, pts_k, dirs_k = ops.occupancy_compact(pd, gd, 32, dirs.to(dev), dirs_div)
    assert torch.equal(kept, want)
    assert torch.equal(pts_k, pd[want.long()])
    assert torch.equal(dirs_k, dirs.to(dev)[(want.long() // dirs_div)])
    back = torch.full((P,), -1, dtype=torch.int32, device=dev)
    back[want.long()] = torch.arange(want.numel(), dtype=torch.int32, device=dev)
    assert torch.equal(slot, back)
    for g in (torch.zeros_like(gd), torch.ones_like(gd)):       # nothing kept, everything kept
        kept, slot, pts_k, _ = ops.occupancy_compact(pd, g, 32)
        assert kept.numel() == (0 if int(g[0]) == 0 else P) and pts_k.shape == (kept.numel(), 3)


def _ngp_with_structure(dev, seed=2):
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    torch.manual_seed(seed)
    ref = O.FastNGPNeRFOracle(log2_hashmap_size=14)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k != "fc_density.2.bias":
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() > 1 else 0.1))
        ref.fc_density[2].bias[0] = -1.0
    m = FastNGPNeRF(log2_hashmap_size=14)
    m.load_state_dict(ref.state_dict())
    return m.to(dev)


def _rays(n_side, dev):
    from nerf_amd.Camera import Camera, look_at_camera
    eye = torch.tensor([0.9, 0.5, 0.6])
    cam = Camera(None, None, (30.0, 30.0), n_side, n_side, c2w=look_at_camera(eye, 2 * eye, torch.tensor([0.0, 0.0, 1.0])).to(dev))
    o, d = cam.get_rays()
    return o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()


def test_hash_grid_training_render_skipping_equals_masking(dev):
    from nerf_amd.render import get_density_grid, ngp_loss, render_rays
    m = _ngp_with_structure(dev)
    grid = get_density_grid(m, dev, 32)
    occupied = float(grid.float().mean())
    assert 0.05 < occupied < 0.95, occupied
    o, d = _rays(24, dev)
    R, N = o.shape[0], 64
    g = torch.Generator().manual_seed(3)
    u, gt = torch.rand(R, N, generator=g).to(dev), torch.rand(R, 3, generator=g).to(dev)
    grads = {}
    outs = {}
    for skip in (False, True):
        for p in m.parameters():
            p.grad = None
        out = render_rays(m, o, d, 0.05, 1.8, N, t_rand=u, density_grid=grid, grid_res=32, skip_empty=skip)
        ngp_loss(out, gt).backward()
        outs[skip] = {k: v.detach().clone() for k, v in out.items()}
        grads[skip] = [p.grad.clone() for p in m.parameters()]
    dropped = float((outs[True]["weights"] == 0).float().mean())
    assert 0.05 < dropped < 0.98, dropped
    for k in ("rgb", "weights", "depth", "z_vals"):
        assert torch.equal(outs[True][k], outs[False][k]), k         # bit-identical render
    for (n, _), a, b in zip(m.named_parameters(), grads[True], grads[False]):
        assert float(b.abs().max()) > 0, n
        assert rel_err(a, b) < 1e-5, (n, rel_err(a, b))             # the same terms, summed in another order


def test_vanilla_hierarchical_render_skipping_equals_masking(dev):
    """Coarse + fine vanilla MLPs with an occupancy grid: both passes skip; also the inference renderer."""
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import nerf_loss, render_image, render_rays
    torch.manual_seed(0)
    coarse, fine = NeRF().to(dev), NeRF().to(dev)
    grid = _ball_grid(32, 0.6).to(dev)
    o, d = _rays(20, dev)
    R = o.shape[0]
    g = torch.Generator().manual_seed(4)
    u, uf, gt = torch.rand(R, 64, generator=g).to(dev), torch.rand(R, 128, generator=g).to(dev), torch.rand(R, 3, generator=g).to(dev)
    params = list(coarse.parameters()) + list(fine.parameters())
    res = {}
    for skip in (False, True):
        for p in params:
            p.grad = None
        out = render_rays(coarse, o, d, 0.05, 1.8, 64, t_rand=u, white_bkgd=True, fine_model=fine, n_importance=128, u_fine=uf,
                          density_grid=grid, grid_res=32, skip_empty=skip)
        nerf_loss(out, gt).backward()
        res[skip] = ({k: v.detach().clone() for k, v in out.items()}, [p.grad.clone() for p in params])
    for k in res[False][0]:
        assert torch.equal(res[True][0][k], res[False][0][k]), k
    for a, b in zip(res[True][1], res[False][1]):
        assert rel_err(a, b) < 1e-5
    img = {}
    for skip in (False, True):
        img[skip] = render_image(coarse, o.view(20, 20, 3), d.view(20, 20, 3), 0.05, 1.8, 64, grid_res=32, density_grid=grid,
                                 fine_model=fine, n_importance=64, rays_per_tile=150, return_depth=True, skip_empty=skip)
    assert torch.equal(img[True][0], img[False][0]) and torch.equal(img[True][1], img[False][1])


@pytest.mark.parametrize("kind", ["ngp", "vanilla"])
def test_render_with_nothing_occupied(dev, kind):
    """An empty grid keeps no sample: the model runs on zero points, the render is the background and the step still
    goes through (the reference's masked render of the same grid gives the same picture)."""
    from nerf_amd.render import nerf_loss, render_image, render_rays
    if kind == "ngp":
        m, fine, kw = _ngp_with_structure(dev), None, {}
    else:
        from nerf_amd.models.NeRF import NeRF
        torch.manual_seed(0)
        m, fine = NeRF().to(dev), NeRF().to(dev)
        kw = dict(fine_model=fine, n_importance=16, white_bkgd=True)
    o, d = _rays(12, dev)
    grid = torch.zeros(32 ** 3, dtype=torch.uint8, device=dev)
    u = torch.rand(o.shape[0], 24, generator=torch.Generator().manual_seed(1)).to(dev)
    outs = {}
    for skip in (False, True):
        for p in list(m.parameters()) + (list(fine.parameters()) if fine else []):
            p.grad = None
        out = render_rays(m, o, d, 0.05, 1.8, 24, t_rand=u, density_grid=grid, grid_res=32, skip_empty=skip, **kw)
        nerf_loss(out, torch.rand(o.shape[0], 3, generator=torch.Generator().manual_seed(2)).to(dev)).backward()
        outs[skip] = out
        for p in m.parameters():
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
    background = 1.0 if kind == "vanilla" else 0.0
    for key in ("rgb", "weights", "depth") + (("rgb_fine",) if fine else ()):
        assert torch.equal(outs[True][key], outs[False][key]), key
    assert float((outs[True]["rgb"].detach() - background).abs().max()) == 0.0 and float(outs[True]["weights"].detach().abs().max()) == 0.0
    img = render_image(m, o.view(12, 12, 3), d.view(12, 12, 3), 0.05, 1.8, n_samples=24, grid_res=32, density_grid=grid,
                       skip_empty=True, white_bkgd=(kind == "vanilla"))
    assert float((img - background).abs().max()) == 0.0


def test_skipping_with_the_count_left_on_the_device(dev):
    """``skip_empty="device"``: no read-back, full-size tensors, the kernels take the number of kept samples from device
    memory.  The render equals the masked one bit for bit, the gradients up to summation order -- eagerly and replayed
    from a HIP graph with other batches copied in (the count changes from replay to replay)."""
    from nerf_amd.dist import GradBuckets
    from nerf_amd.graph import GraphedStep
    from nerf_amd.optim import FusedAdam
    from nerf_amd.render import ngp_loss, render_rays, train_step
    m = _ngp_with_structure(dev)
    o, d = _rays(40, dev)                      # 1600 rays x 48 samples = 76,800 points: the binned scatter
    N = 48
    grid = _ball_grid(32, 0.55).to(dev)
    gen = torch.Generator().manual_seed(3)
    u = torch.rand(o.shape[0], N, generator=gen).to(dev)
    gt = torch.rand(o.shape[0], 3, generator=gen).to(dev)
    res = {}
    for mode in (False, True, "device"):
        for p in m.parameters():
            p.grad = None
        out = render_rays(m, o, d, 0.05, 1.8, N, t_rand=u, density_grid=grid, grid_res=32, skip_empty=mode)
        ngp_loss(out, gt).backward()
        res[mode] = (out, {k: p.grad.clone() for k, p in m.named_parameters()})
    for key in ("rgb", "weights", "depth"):
        assert torch.equal(res["device"][0][key], res[False][0][key]), key
    for k, g in res[False][1].items():
        assert float(g.abs().max()) > 0, k
        assert rel_err(res["device"][1][k], g) < 1e-5, (k, rel_err(res["device"][1][k], g))
        assert rel_err(res["device"][1][k], res[True][1][k]) < 1e-5, k
    # an empty and a full grid through the same path
    for g_all, want in ((torch.zeros_like(grid), 0.0), (torch.ones_like(grid), None)):
        out = render_rays(m, o, d, 0.05, 1.8, N, t_rand=u, density_grid=g_all, grid_res=32, skip_empty="device")
        ref = render_rays(m, o, d, 0.05, 1.8, N, t_rand=u, density_grid=g_all, grid_res=32)
        assert torch.equal(out["rgb"], ref["rgb"]) and (want is None or float(out["rgb"].detach().abs().max()) == want)

    # captured: lr 0 so that every replay sees the same parameters; the rays (and with them the count) change per replay
    buckets = GradBuckets([list(m.hash_encoder.parameters()), [p for n, p in m.named_parameters() if not n.startswith("hash_encoder")]])
    opt = FusedAdam(buckets.parameters(), lr=0.0, eps=1e-15, capturable=True)
    s_o, s_d = o.clone(), d.clone()
    z_lin = torch.linspace(0.05, 1.8, N).to(dev)

    def step(mode):
        loss = train_step(m, s_o, s_d, gt, 0.05, 1.8, N, loss_fn=ngp_loss, buckets=buckets, t_rand=u, z_lin=z_lin,
                          density_grid=grid, grid_res=32, skip_empty=mode)
        opt.step()
        return loss

    g = GraphedStep(lambda: step("device"), warmup=2, optimizers=[opt])
    for shift in (0.0, 0.3, -0.2):
        s_o.copy_(o + torch.tensor([shift, 0.0, 0.0], device=dev))
        torch.cuda.synchronize()
        loss_g = float(g())
        got = [f.clone() for f in buckets.flats]
        loss_e = float(step(False))
        for a, b in zip(got, buckets.flats):
            assert rel_err(a, b) < 1e-5, (shift, rel_err(a, b))
        assert abs(loss_g - loss_e) <= 1e-6 * max(1.0, abs(loss_e)), (shift, loss_g, loss_e)
