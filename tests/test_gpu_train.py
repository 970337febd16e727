"""End-to-end parity: a whole render + loss + backward (coarse and hierarchical
fine pass) against the CPU oracle, and PSNR after a fixed number of training
steps on the analytic scene."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden, rel_err
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def _rays(n_side=12, f=30.0):
    c2w = O.look_at_camera(torch.tensor([2.5, 1.5, 1.0]), torch.tensor([5.0, 3.0, 2.0]), torch.tensor([0.0, 0.0, 1.0]))
    o, d = O.get_rays(c2w, (f, f), n_side, n_side)
    return o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()


def test_hierarchical_render_step_matches_oracle(dev):
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import nerf_loss, render_rays
    torch.manual_seed(0)
    ref_c, ref_f = O.NeRFOracle(), O.NeRFOracle()
    coarse, fine = NeRF(), NeRF()
    coarse.load_state_dict(ref_c.state_dict())
    fine.load_state_dict(ref_f.state_dict())
    coarse, fine = coarse.to(dev), fine.to(dev)
    o, d = _rays()
    R, N, NF = o.shape[0], 64, 128
    g = torch.Generator().manual_seed(1)
    u, uf, gt = torch.rand(R, N, generator=g), torch.rand(R, NF, generator=g), torch.rand(R, 3, generator=g)
    out = render_rays(coarse, o.to(dev), d.to(dev), 2.0, 6.0, N, t_rand=u.to(dev), white_bkgd=True, fine_model=fine,
                      n_importance=NF, u_fine=uf.to(dev))
    nerf_loss(out, gt.to(dev)).backward()
    c_map, w, z, (f_map, w_f, z_all) = O.render_rays(ref_c, o, d, 2.0, 6.0, N, t_rand=u, white_bkgd=True,
                                                      fine_model=ref_f, u_fine=uf)
    (torch.nn.functional.mse_loss(c_map, gt) + torch.nn.functional.mse_loss(f_map, gt)).backward()
    assert torch.equal(out["z_vals"].cpu(), z)                        # stratified depths: bit-exact
    # contract: colour within 1e-3 absolute (observed ~1e-6)
    assert float((out["rgb"].detach().cpu() - c_map.detach()).abs().max()) < 1e-4
    assert float((out["rgb_fine"].detach().cpu() - f_map.detach()).abs().max()) < 1e-4
    # the fine depths depend on the coarse weights (1e-6 apart): compare, do not require bit equality
    assert float((out["z_fine"].cpu() - z_all).abs().max()) < 1e-3
    assert float((out["depth_fine"].detach().cpu() - O.depth(w_f, z_all).detach()).abs().max()) < 1e-3
    # Gradients.  In a render the gradient mass sits on the few samples with non-zero
    # weight, so one ReLU whose pre-activation is within an ulp of zero (and therefore
    # switches between two float32 implementations) moves a trunk gradient by ~1e-3.
    # Measured against a float64 evaluation of the same pipeline, over several seeds: the
    # float32 CPU path 1.5e-3, the exact-f32 MFMA kernels 3e-4 / 1.9e-3 / 3e-2, the
    # split-operand MFMA kernels 1.5e-3 / 1.9e-3 / 3e-2 (maximum over parameters; medians 1e-6..5e-5),
    # and the CPU result differs between CPUs.  The 1e-3 contract is therefore held where
    # it is well-posed -- per kernel, with dense upstream gradients (test_gpu_mlp_vanilla,
    # 1e-4) -- and the whole render is held to 5e-3 against both the float64 evaluation at
    # the GPU's own sample positions and the float32 CPU path, with the median at 1e-4.
    import copy
    c64, f64 = copy.deepcopy(ref_c).double(), copy.deepcopy(ref_f).double()
    c64.zero_grad()
    f64.zero_grad()
    vd = (d / d.norm(dim=-1, keepdim=True)).double()

    def eval64(model, zz):
        n = zz.shape[1]
        pts = (o[:, None, :] + d[:, None, :] * zz[..., None]).double()   # float32 points, as every path forms them
        rgb, sig = model(pts.reshape(-1, 3), vd[:, None, :].expand(R, n, 3).reshape(-1, 3))
        return O.volume_render(rgb.view(R, n, 3), sig.view(R, n), zz.double(), True)[0]

    loss64 = torch.nn.functional.mse_loss(eval64(c64, out["z_vals"].cpu()), gt.double()) + \
        torch.nn.functional.mse_loss(eval64(f64, out["z_fine"].cpu()), gt.double())
    loss64.backward()
    errs = []
    for (k, p), q, q64 in zip(list(coarse.named_parameters()) + list(fine.named_parameters()),
                              list(ref_c.parameters()) + list(ref_f.parameters()),
                              list(c64.parameters()) + list(f64.parameters())):
        errs.append(rel_err(p.grad.cpu(), q64.grad))
        assert errs[-1] < 5e-3, k
        assert rel_err(p.grad.cpu(), q.grad) < 5e-3, k
    assert sorted(errs)[len(errs) // 2] < 1e-4


def test_ngp_render_step_with_weight_regulariser(dev):
    """The hash-grid training loss reads both outputs of the compositor
    (src/train_nerf_ngp.py:221-222)."""
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.render import render_rays
    torch.manual_seed(2)
    ref = O.FastNGPNeRFOracle(log2_hashmap_size=14)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.startswith("hash_encoder"):
                p.mul_(300.0)
            elif k != "fc_density.2.bias":
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() > 1 else 0.1))
    m = FastNGPNeRF(log2_hashmap_size=14)
    m.load_state_dict(ref.state_dict())
    m = m.to(dev)
    o, d = _rays()
    o = o * 0.1                                   # unit-sphere scale, as the capture dataset normalises poses
    R, N = o.shape[0], 64
    g = torch.Generator().manual_seed(3)
    u, gt = torch.rand(R, N, generator=g), torch.rand(R, 3, generator=g)
    out = render_rays(m, o.to(dev), d.to(dev), 0.05, 1.2, N, t_rand=u.to(dev))
    loss = torch.nn.functional.mse_loss(out["rgb"], gt.to(dev)) + 0.05 * (1 - out["weights"].sum(dim=-1)).abs().mean()
    loss.backward()
    c_map, w, z, _ = O.render_rays(ref, o, d, 0.05, 1.2, N, t_rand=u, chunk=16384)
    loss_o = torch.nn.functional.mse_loss(c_map, gt) + 0.05 * (1 - w.sum(dim=-1)).abs().mean()
    loss_o.backward()
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 1e-5
    assert float((out["rgb"].detach().cpu() - c_map.detach()).abs().max()) < 1e-4
    for (k, p), q in zip(m.named_parameters(), ref.parameters()):
        err = float((p.grad.cpu() - q.grad).norm())
        assert err <= 1e-3 * float(q.grad.norm()) + 1e-9, (k, err, float(q.grad.norm()))


def test_psnr_after_fixed_steps_matches_cpu_training(dev):
    """Train the same model on the analytic torus+sphere scene for a fixed number
    of steps on the GPU and with the CPU oracle (same init, same views, same
    uniforms): PSNR on a held-out view must agree within 0.1 dB."""
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import render_rays
    g7 = load_golden("g7_scene.npz")
    H, W, f, near, far, n = g7["cfg"]
    H, W, n = int(H), int(W), int(n)
    poses, imgs = T(g7["c2w"]), T(g7["img"])
    views = list(range(poses.shape[0] - 1))
    held = poses.shape[0] - 1
    steps = 40
    torch.manual_seed(0)
    ref = O.NeRFOracle()
    m = NeRF()
    m.load_state_dict(ref.state_dict())
    m = m.to(dev)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=5e-4)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    gen = torch.Generator().manual_seed(9)
    for s in range(steps):
        v = views[s % len(views)]
        o, d = O.get_rays(poses[v], (float(f), float(f)), H, W)
        o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
        u = torch.rand(H * W, n, generator=gen)
        gt = imgs[v].reshape(-1, 3)
        c_map, _, _, _ = O.render_rays(ref, o, d, float(near), float(far), n, t_rand=u)
        loss_ref = torch.nn.functional.mse_loss(c_map, gt)
        opt_ref.zero_grad()
        loss_ref.backward()
        opt_ref.step()
        out = render_rays(m, o.to(dev), d.to(dev), float(near), float(far), n, t_rand=u.to(dev))
        loss = torch.nn.functional.mse_loss(out["rgb"], gt.to(dev))
        opt.zero_grad()
        loss.backward()
        opt.step()
    o, d = O.get_rays(poses[held], (float(f), float(f)), H, W)
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    with torch.no_grad():
        img_ref = O.render_rays(ref, o, d, float(near), float(far), n, t_rand=None)[0]
        img = render_rays(m, o.to(dev), d.to(dev), float(near), float(far), n, perturb=False)["rgb"].cpu()
    gt = imgs[held].reshape(-1, 3)
    psnr_ref, psnr = float(O.psnr(img_ref, gt)), float(O.psnr(img, gt))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-3 * max(1e-3, float(loss_ref.detach())) + 1e-5
    assert abs(psnr - psnr_ref) < 0.1, (psnr, psnr_ref)
    assert psnr > float(O.psnr(torch.zeros_like(gt), gt)) - 3.0   # training did not diverge


# --------------------------------------------------------------------------------------------------
# The 1e-3 gradient contract in a well-posed form.  A whole render puts its gradient mass on a few samples,
# so a ReLU whose pre-activation is within rounding of zero -- open in one float32 implementation, shut in
# another -- moves a trunk gradient by more than 1e-3.  The forward pass records which ReLUs it found open;
# the oracle is evaluated on exactly that piecewise-linear branch, and the few disagreements with the
# oracle's own pattern are checked to be such rounding cases.
# --------------------------------------------------------------------------------------------------
def _forced_mask_parity(dev, seed, o, d, near, far, N, NF, ndc=None, white=True):
    import copy
    from nerf_amd import ops
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import nerf_loss, render_rays
    torch.manual_seed(seed)
    ref_c, ref_f = O.NeRFOracle(), O.NeRFOracle()
    coarse, fine = NeRF(), NeRF()
    coarse.load_state_dict(ref_c.state_dict())
    fine.load_state_dict(ref_f.state_dict())
    coarse, fine = coarse.to(dev), fine.to(dev)
    R = o.shape[0]
    g = torch.Generator().manual_seed(seed + 1)
    u, uf, gt = torch.rand(R, N, generator=g), torch.rand(R, NF, generator=g), torch.rand(R, 3, generator=g)
    out = render_rays(coarse, o.to(dev), d.to(dev), near, far, N, t_rand=u.to(dev), white_bkgd=white, fine_model=fine,
                      n_importance=NF, u_fine=uf.to(dev), ndc=ndc)
    nerf_loss(out, gt.to(dev)).backward()
    # the ReLU patterns of the two forward passes, from a second forward over the same points (deterministic)
    od, dd = o.to(dev), d.to(dev)
    vd = dd / dd.norm(dim=-1, keepdim=True)
    if ndc is not None:
        od, dd = ops.ndc_rays(ndc[0], ndc[1], ndc[2], ndc[3], od, dd)
    masks = []
    for model, zz in ((coarse, out["z_vals"]), (fine, out["z_fine"])):
        pts = (od[:, None, :] + dd[:, None, :] * zz[..., None]).reshape(-1, 3)   # what the samplers write
        packed = ops.vanilla_pack(model._params(), False)
        rgb, sigma, ws = ops.vanilla_fwd_raw(pts.contiguous(), vd.contiguous(), zz.shape[1], packed, True)
        masks.append(ops.vanilla_relu_masks(ws, pts.shape[0]).cpu())
    c_map, w, z, (f_map, w_f, z_all) = O.render_rays(ref_c, o, d, near, far, N, t_rand=u, white_bkgd=white, fine_model=ref_f,
                                                      u_fine=uf, ndc=ndc, masks=masks[0], masks_fine=masks[1],
                                                      z_fixed=out["z_fine"].cpu(), chunk=1 << 16)
    (torch.nn.functional.mse_loss(c_map, gt) + torch.nn.functional.mse_loss(f_map, gt)).backward()
    assert torch.equal(out["z_vals"].cpu(), z)
    assert float((out["rgb"].detach().cpu() - c_map.detach()).abs().max()) < 1e-4
    assert float((out["rgb_fine"].detach().cpu() - f_map.detach()).abs().max()) < 1e-4
    worst = 0.0
    for (k, p), q in zip(list(coarse.named_parameters()) + list(fine.named_parameters()),
                         list(ref_c.parameters()) + list(ref_f.parameters())):
        e = rel_err(p.grad.cpu(), q.grad)
        assert e < 1e-3, (seed, k, e)                       # the contract; observed ~1e-5
        worst = max(worst, e)
    # disagreements between the GPU's pattern and a float64 evaluation's: few, and all of them rounding cases
    flips = total = 0
    for model, zz, m in ((ref_c, out["z_vals"].cpu(), masks[0]), (ref_f, out["z_fine"].cpu(), masks[1])):
        m64 = copy.deepcopy(model).double()
        oo, d2 = (o, d) if ndc is None else O.ndc_rays(ndc[0], ndc[1], ndc[2], ndc[3], o, d)
        pts = (oo[:, None, :] + d2[:, None, :] * zz[..., None]).reshape(-1, 3).double()
        dirs = (d / d.norm(dim=-1, keepdim=True))[:, None, :].expand(R, zz.shape[1], 3).reshape(-1, 3).double()
        rec = []
        with torch.no_grad():
            m64(pts, dirs, masks=m.double(), record=rec)    # pre-activations along the GPU's own branch
        pre = torch.cat(rec, dim=1)
        differ = (pre > 0) != m.bool()
        flips += int(differ.sum())
        total += differ.numel()
        if differ.any():
            assert float(pre[differ].abs().max()) < 2e-5, "a ReLU decision differs away from zero"
    assert flips <= 1e-4 * total
    return worst, flips, total


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_render_gradients_meet_1e3_on_the_forward_branch(dev, seed):
    o, d = _rays()
    worst, flips, total = _forced_mask_parity(dev, seed, o, d, 2.0, 6.0, 64, 128)
    print(f"seed {seed}: worst relative gradient error {worst:.2e}; {flips} of {total} ReLU decisions differ from float64")


# --------------------------------------------------------------------------------------------------
# Forward-facing scenes in normalised device coordinates (not in the reference: parity unpinned, pinned
# to the oracle's definition): rays -> NDC -> depths in [0,1] -> hierarchical -> composite.
# --------------------------------------------------------------------------------------------------
def _forward_facing_pose():
    """A camera near the origin looking along -z with a small tilt (what an LLFF capture looks like
    after its poses are recentred)."""
    rot = torch.linalg.matrix_exp(torch.tensor([[0.0, -0.03, 0.05], [0.03, 0.0, -0.02], [-0.05, 0.02, 0.0]]))
    return torch.cat([rot, torch.tensor([[0.12], [-0.07], [0.05]])], dim=1)


@pytest.mark.parametrize("which", ["golden_llff_pose", "forward_facing"])
def test_ndc_hierarchical_render_step_matches_oracle(dev, which):
    g1 = load_golden("g1_raygen.npz")
    H, W = [int(v) // 3 for v in g1["llff_hw"]]              # 21 x 28 rays
    f = float(g1["llff_f"][0]) / 3
    c2w = T(g1["llff_c2w"]) if which == "golden_llff_pose" else _forward_facing_pose()
    o, d = O.get_rays(c2w, (f, f), H, W)
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    worst, flips, total = _forced_mask_parity(dev, 4, o, d, 0.0, 1.0, 64, 128, ndc=(H, W, f, 1.0), white=False)
    print(f"ndc {which}: worst relative gradient error {worst:.2e}; {flips} ReLU decisions of {total} differ")


def test_ndc_full_image_step_504x378(dev):
    """The whole forward-facing configuration (504 x 378 rays, 64 + 128 samples, coarse and fine 8x256 MLPs)
    as ONE chunked training step, and the properties that hold at any size: depths inside [0,1] and sorted,
    weights a sub-probability, expected depth inside [0,1], finite gradients on every parameter."""
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import render_rays, train_step
    from nerf_amd.Camera import Camera
    H, W, f = 378, 504, 3321.11 / 8
    cam = Camera(None, None, (f, f), H, W, c2w=_forward_facing_pose().to(dev))
    o, d = cam.get_rays()
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    torch.manual_seed(5)
    coarse, fine = NeRF().to(dev), NeRF().to(dev)
    R = H * W
    gt = torch.rand(R, 3, device=dev)
    kw = dict(fine_model=fine, n_importance=128, ndc=(H, W, f, 1.0))
    for p in list(coarse.parameters()) + list(fine.parameters()):
        p.grad = None
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    loss = train_step(coarse, o, d, gt, 0.0, 1.0, 64, **kw)
    assert torch.isfinite(loss)
    assert torch.cuda.max_memory_allocated(dev) - base < 40 << 30       # 190,512 rays unchunked would need 860 GB
    for p in list(coarse.parameters()) + list(fine.parameters()):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0
    with torch.no_grad():
        sel = torch.arange(0, R, 7, device=dev)
        out = render_rays(coarse, o[sel], d[sel], 0.0, 1.0, 64, **kw)
    for zk, wk, dk in (("z_vals", "weights", "depth"), ("z_fine", "weights_fine", "depth_fine")):
        z, w, dep = out[zk], out[wk], out[dk]
        assert float(z.min()) >= 0.0 and float(z.max()) <= 1.0
        assert bool((z[:, 1:] >= z[:, :-1]).all())
        assert float(w.min()) >= 0.0 and float(w.sum(dim=-1).max()) <= 1.0 + 1e-5
        assert float(dep.min()) >= 0.0 and float(dep.max()) <= 1.0 + 1e-5
    assert out["z_fine"].shape[1] == 192


# --------------------------------------------------------------------------------------------------
# Ray-chunked training step: bounded activation memory, same gradients.
# --------------------------------------------------------------------------------------------------
def _lego_batch(dev, R, seed=0):
    from nerf_amd.Camera import Camera, look_at_camera
    eye = torch.tensor([3.2, 1.8, 1.7])
    c2w = look_at_camera(eye, 2 * eye, torch.tensor([0.0, 0.0, 1.0]))
    g = torch.Generator().manual_seed(seed)
    pix = torch.randperm(800 * 800, generator=g)[:R].to(torch.int32)
    cam = Camera(None, None, (1111.11, 1111.11), 800, 800, c2w=c2w.to(dev))
    o, d = cam.get_ray_batch(pix.to(dev))
    return o, d, torch.rand(R, 64, generator=g).to(dev), torch.rand(R, 128, generator=g).to(dev), torch.rand(R, 3, generator=g).to(dev)


def test_chunked_step_equals_unchunked_step(dev):
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import nerf_loss, render_rays, train_step
    R = 8192
    o, d, u, uf, gt = _lego_batch(dev, R)
    torch.manual_seed(0)
    coarse, fine = NeRF().to(dev), NeRF().to(dev)
    params = list(coarse.parameters()) + list(fine.parameters())
    out = render_rays(coarse, o, d, 2.0, 6.0, 64, t_rand=u, white_bkgd=True, fine_model=fine, n_importance=128, u_fine=uf)
    loss_ref = nerf_loss(out, gt)
    loss_ref.backward()
    want = [p.grad.clone() for p in params]
    del out
    for chunk in (1024, 3000):                              # 8 equal chunks; 3 chunks with a ragged tail
        for p in params:
            p.grad = None
        loss = train_step(coarse, o, d, gt, 2.0, 6.0, 64, chunk_rays=chunk, t_rand=u, u_fine=uf, white_bkgd=True,
                          fine_model=fine, n_importance=128)
        assert abs(float(loss) - float(loss_ref)) < 1e-6
        for (k, p), g0 in zip(list(coarse.named_parameters()) + list(fine.named_parameters()), want):
            # Same terms, another summation order: 1e-5.  A chunk size that is no power of two also scales the
            # loss by a factor that is none, so every term moves by an ulp; the density head's gradient is a sum
            # of large terms that cancel to 1e-3..1e-4 of their size (a uniform shift of log-density barely
            # changes a render), and such a sum follows its terms only to ~1e-4 (measured 2e-4..4e-4, the same
            # from run to run).
            tol = 2e-3 if k.startswith("sigma_out") and chunk & (chunk - 1) else 1e-5
            assert rel_err(p.grad, g0) < tol, (chunk, k)


def test_65536_ray_step_fits_in_bounded_memory(dev):
    """The data-parallel configuration's per-GPU batch: 65,536 rays x (64 + 192) evaluations of an 8x256 MLP keep
    20 KB each for backward, 300 GB if the step ran as one piece."""
    from nerf_amd.dist import GradBuckets
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import train_step
    R = 65536
    o, d, u, uf, gt = _lego_batch(dev, R, seed=1)
    torch.manual_seed(0)
    coarse, fine = NeRF().to(dev), NeRF().to(dev)
    buckets = GradBuckets([list(fine.parameters()), list(coarse.parameters())])
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    loss = train_step(coarse, o, d, gt, 2.0, 6.0, 64, t_rand=u, u_fine=uf, white_bkgd=True, fine_model=fine,
                      n_importance=128, buckets=buckets)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(dev) - base
    assert torch.isfinite(loss) and 0.0 < float(loss) < 2.0
    assert peak < 40 << 30, peak                            # default workspace budget: 32 GiB + outputs
    for p in buckets.parameters():
        assert bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0
    assert coarse.layers_xyz[0].weight.grad.data_ptr() == buckets.views[1][0].data_ptr()


def test_inference_under_no_grad_keeps_no_workspace(dev):
    """Parameters that require grad do not make a no_grad forward save activations (10 KB per point)."""
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.models.NeRF import NeRF
    torch.manual_seed(0)
    P = 1 << 20
    x = torch.rand(P, 3, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
    # NeRF: outputs 16 MB + packed weights; hash-grid model: + the 128 B/point encoding between its two launches
    # (a training forward would keep 10.4 GB and 0.96 GB of activations)
    for model, limit in ((NeRF().to(dev), 64 << 20), (FastNGPNeRF(log2_hashmap_size=14).to(dev), 192 << 20)):
        assert all(p.requires_grad for p in model.parameters())
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        with torch.no_grad():
            rgb, sigma = model(x, dirs)
        torch.cuda.synchronize()
        assert torch.cuda.max_memory_allocated(dev) - base < limit, type(model).__name__
        assert not rgb.requires_grad
        rgb2, _ = model(x[:4096], dirs[:4096])
        assert rgb2.requires_grad and torch.equal(rgb2.detach(), rgb[:4096])


def test_ngp_training_render_with_occupancy_grid(dev):
    """sigma zeroed outside the occupancy grid before compositing, masked samples without gradient
    (src/train_nerf_ngp.py:113-117), against the oracle composed the same way."""
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.render import get_density_grid, render_rays
    torch.manual_seed(2)
    ref = O.FastNGPNeRFOracle(log2_hashmap_size=14)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k != "fc_density.2.bias":                     # tables too: the density must vary over space
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() > 1 else 0.1))
        ref.fc_density[2].bias[0] = -1.0                     # densities on both sides of the threshold of 1
    m = FastNGPNeRF(log2_hashmap_size=14)
    m.load_state_dict(ref.state_dict())
    m = m.to(dev)
    grid_res = 32
    grid = get_density_grid(m, dev, grid_res)
    grid_ref = O.get_density_grid(ref, grid_res)
    occupied = float(grid_ref.float().mean())
    assert 0.05 < occupied < 0.95, occupied                  # a grid that masks something and keeps something
    # cells whose density is within rounding of the threshold may differ; use one grid on both sides
    assert float((grid.cpu() != grid_ref).float().mean()) < 1e-3
    o, d = _rays()
    o = o * 0.1
    R, N = o.shape[0], 64
    g = torch.Generator().manual_seed(3)
    u, gt = torch.rand(R, N, generator=g), torch.rand(R, 3, generator=g)
    out = render_rays(m, o.to(dev), d.to(dev), 0.05, 1.2, N, t_rand=u.to(dev), density_grid=grid_ref.to(dev), grid_res=grid_res)
    loss = torch.nn.functional.mse_loss(out["rgb"], gt.to(dev)) + 0.05 * (1 - out["weights"].sum(dim=-1)).abs().mean()
    loss.backward()
    c_map, w, z, _ = O.render_rays(ref, o, d, 0.05, 1.2, N, t_rand=u, chunk=16384, density_grid=grid_ref, grid_res=grid_res)
    loss_o = torch.nn.functional.mse_loss(c_map, gt) + 0.05 * (1 - w.sum(dim=-1)).abs().mean()
    loss_o.backward()
    masked = float((w == 0).float().mean())
    assert 0.02 < masked < 0.98, masked
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 1e-5
    assert float((out["rgb"].detach().cpu() - c_map.detach()).abs().max()) < 1e-4
    assert float((out["weights"].detach().cpu() - w.detach()).abs().max()) < 1e-5
    for (k, p), q in zip(m.named_parameters(), ref.parameters()):
        err = float((p.grad.cpu() - q.grad).norm())
        assert err <= 1e-3 * float(q.grad.norm()) + 1e-9, (k, err, float(q.grad.norm()))


def test_psnr_after_320_steps_matches_the_reference_trajectory(dev):
    """The reference's own pieces (its NeRF, Camera sampling and compositing, torch Adam and clip_grad_norm_) were
    trained for 320 steps on the analytic torus + sphere scene at the size of its synthetic trainer (100 x 100
    rays, 64 samples, 40 views; oracle/gen_golden.py:g9) and the losses and held-out PSNRs recorded.  The same
    run on the GPU -- same seeded init, same uniforms, fused Adam and the fused gradient clipping -- must stay
    within 0.1 dB of it in the PSNR the reference logs: the mean over an epoch's views (src/train_nerf.py:175-190),
    for every one of the 8 epochs.  (The PSNR of ONE held-out view at ONE instant jumps by a dB between marks
    this early in training, and two float32 trajectories 1e-3 apart in loss land 0.1 dB apart on it.)"""
    from nerf_amd import ops
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.optim import FusedAdam
    from nerf_amd.render import psnr, render_rays
    g = load_golden("g9_train.npz")
    H, W, f, near, far, n, boost = [float(v) for v in g["cfg"]]
    H, W, n = int(H), int(W), int(n)
    poses = T(g["c2w"])
    rays, imgs = [], []
    for k in range(poses.shape[0]):                        # ground truth: the analytic field through the oracle
        o, d = O.get_rays(poses[k], (f, f), H, W)
        pts, z = O.sample_points_along_rays(o, d, near, far, n, None)
        rgb, sigma = O.analytic_field(pts)
        img, _ = O.volume_render(rgb, sigma * boost, z)
        assert abs(float(img.double().sum()) - float(g["img_sum"][k])) < 1e-6 * max(1.0, float(g["img_sum"][k]))
        # exp and sqrt round differently from one CPU to another: the last bit, not more
        assert float((img[::10, ::10] - T(g["img_sample"][k])).abs().max()) < 1e-6
        rays.append((o.reshape(-1, 3).contiguous().to(dev), d.reshape(-1, 3).contiguous().to(dev)))
        imgs.append(img.reshape(-1, 3).to(dev))
    torch.manual_seed(0)
    m = NeRF().to(dev)
    params = list(m.parameters())
    opt = FusedAdam(params, lr=5e-4)
    want_loss, marks, want_psnr = g["loss"], [int(s) for s in g["psnr_steps"]], g["psnr"]
    steps = len(want_loss)
    losses, psnrs = [], []
    for step in range(steps):
        v = step % 40
        torch.manual_seed(1000 + step)
        u = torch.rand(H, W, n).reshape(H * W, n).to(dev)   # what Camera.sample_points_along_rays drew with rand_like
        out = render_rays(m, rays[v][0], rays[v][1], near, far, n, t_rand=u)
        loss = torch.nn.functional.mse_loss(out["rgb"], imgs[v])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        ops.clip_grad_norm_(params, 1.0)
        opt.step()
        losses.append(loss.detach())
        if step + 1 in marks:
            with torch.no_grad():
                held = render_rays(m, rays[40][0], rays[40][1], near, far, n, perturb=False)["rgb"]
            psnrs.append(float(psnr(held, imgs[40])))
    losses = torch.stack(losses).cpu().double().numpy()
    print("psnr gpu", ["%.3f" % p for p in psnrs], "reference", ["%.3f" % p for p in want_psnr])
    print("loss gpu/ref at 1, 40, 160, last:", [(float(losses[i]), float(want_loss[i])) for i in (0, 39, 159, steps - 1)])
    # the first steps agree to float32 rounding; later the two float32 trajectories drift apart slowly
    assert abs(losses[0] - want_loss[0]) < 1e-5 * want_loss[0]
    assert float(np.max(np.abs(losses[:20] - want_loss[:20]) / want_loss[:20])) < 1e-3
    epoch = lambda l: (-10.0 * np.log10(l)).reshape(-1, 40).mean(axis=1)   # per-view PSNRs, averaged over the epoch
    ep_gpu, ep_ref = epoch(losses), epoch(want_loss)
    print("epoch psnr gpu", np.round(ep_gpu, 3), "reference", np.round(ep_ref, 3))
    assert abs(ep_gpu[-1] - ep_ref[-1]) < 0.1, (ep_gpu, ep_ref)             # the contract: PSNR after the fixed count
    assert float(np.max(np.abs(ep_gpu[:4] - ep_ref[:4]))) < 0.02, (ep_gpu, ep_ref)   # 160 steps: still the same trajectory
    assert float(np.max(np.abs(ep_gpu - ep_ref))) < 0.2, (ep_gpu, ep_ref)   # in between (observed <= 0.07, varies with atomic order)
    for a, b in zip(psnrs, want_psnr):
        assert abs(a - float(b)) < 0.25, (psnrs, want_psnr)
    assert losses[-1] < 0.5 * losses[0]                     # it trained


def test_fused_render_loss_matches_the_tensor_ops(dev):
    """ops.render_loss (MSE of one or two colour maps + the hash-grid loop's weight regulariser, value and gradient)
    against the same loss written with torch's ops, at ragged sizes and with a scaled upstream gradient."""
    from nerf_amd import ops
    gen = torch.Generator().manual_seed(31)
    for R, N in [(1, 2), (77, 17), (4096, 64), (5000, 192)]:
        base = [torch.rand(R, 3, generator=gen), torch.rand(R, 3, generator=gen), torch.rand(R, 3, generator=gen),
                torch.rand(R, N, generator=gen) * (2.2 / N)]
        base[3][0] = 1.0 / N                                   # a ray whose weights sum to 1 (to rounding)
        for with_fine, reg in [(False, 0.0), (True, 0.0), (False, 0.05), (True, 0.3)]:
            rgb, fine, tgt, w = (t.clone().to(dev) for t in base)
            a = [t.clone().requires_grad_(True) for t in (rgb, fine, w)]
            b = [t.clone().requires_grad_(True) for t in (rgb, fine, w)]
            want = torch.nn.functional.mse_loss(a[0], tgt)
            if with_fine:
                want = want + torch.nn.functional.mse_loss(a[1], tgt)
            if reg:
                want = want + reg * (1.0 - a[2].sum(dim=-1)).abs().mean()
            got = ops.render_loss(b[0], tgt, b[1] if with_fine else None, b[2] if reg else None, reg)
            assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)), (R, N, with_fine, reg)
            (want * 0.37).backward()
            (got * 0.37).backward()
            for x, y in zip(a, b):
                if x.grad is None:
                    assert y.grad is None
                    continue
                scale = float(x.grad.abs().max())
                # the regulariser's gradient is a sign: rays whose weights sum to 1 within rounding may take either
                bad = (x.grad - y.grad).abs() > 2e-6 * scale + 1e-12
                assert int(bad.sum()) <= (N if reg else 0), (R, N, with_fine, reg, int(bad.sum()))


def test_one_model_in_both_passes_accumulates_straight_into_the_buckets(dev):
    """fine_model = the coarse model: two calls of the fused op on the same parameters in one backward pass.  With
    buckets the two calls accumulate straight into the same flat buffer (ops.grad_sink); the result must be the sum
    autograd forms without buckets."""
    from nerf_amd.dist import GradBuckets
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.render import train_step
    torch.manual_seed(9)
    model = NeRF().to(dev)
    R = 512
    gen = torch.Generator().manual_seed(10)
    o = torch.tensor([0.0, 0.0, 4.0]).expand(R, 3).contiguous().to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen) * torch.tensor([0.3, 0.3, 0.0]) +
                                      torch.tensor([0.0, 0.0, -1.0]), dim=-1).to(dev)
    gt = torch.rand(R, 3, generator=gen).to(dev)
    kw = dict(t_rand=torch.rand(R, 64, generator=gen).to(dev), u_fine=torch.rand(R, 128, generator=gen).to(dev),
              white_bkgd=True, fine_model=model, n_importance=128)
    loss_ref = train_step(model, o, d, gt, 2.0, 6.0, 64, **kw)
    ref = [p.grad.clone() for p in model.parameters()]
    gb = GradBuckets([list(model.parameters())])
    for chunk in (R, 128):
        loss = train_step(model, o, d, gt, 2.0, 6.0, 64, buckets=gb, chunk_rays=chunk, **kw)
        assert abs(float(loss) - float(loss_ref)) < 1e-6
        assert gb._sent == [True] and gb._pending == [0]
        for p, r in zip(model.parameters(), ref):
            assert p.grad.data_ptr() == gb.views[0][[id(q) for q in gb.buckets[0]].index(id(p))].data_ptr()
            err = float((p.grad - r).double().norm())
            assert err <= 2e-5 * float(r.double().norm()) + 1e-12, (chunk, tuple(p.shape), err)
