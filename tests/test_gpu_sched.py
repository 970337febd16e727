"""Learning-rate schedules over the fused Adam, eagerly and through a replayed HIP graph, against trajectories of the
reference's own pieces under the schedulers its loops use (tests/golden/g10_sched.npz, oracle/gen_golden.py:g10):
ExponentialLR(0.98) per epoch over NeRF() (src/train_nerf.py:86-87,204) and CosineAnnealingLR(T_max=30, eta_min=1e-4)
per epoch over FastNGPNeRF() with Adam(lr=1e-2, eps=1e-15) (src/train_nerf_ngp.py:55,62,375)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _scene(g, tag, dev):
    H, W, f, near, far, n, boost, views, shrink = [float(v) for v in g[tag + "_cfg"]]
    H, W, n, views = int(H), int(W), int(n), int(views)
    rays, imgs = [], []
    for k in range(views):                                  # ground truth: the analytic field through the oracle
        o, d = O.get_rays(T(g[tag + "_c2w"][k]), (f, f), H, W)
        pts, z = O.sample_points_along_rays(o, d, near, far, n, None)
        rgb, sigma = O.analytic_field(pts * shrink)
        img, _ = O.volume_render(rgb, sigma * boost, z)
        want = float(g[tag + "_img_sum"][k])
        assert abs(float(img.double().sum()) - want) < 1e-6 * max(1.0, want)
        rays.append((o.reshape(-1, 3).contiguous().to(dev), d.reshape(-1, 3).contiguous().to(dev)))
        imgs.append(img.reshape(-1, 3).to(dev))
    return dict(H=H, W=W, near=near, far=far, n=n, views=views, rays=rays, imgs=imgs, steps=len(g[tag + "_loss"]))


def _uniforms(sc, seed0, dev):
    """What Camera.sample_points_along_rays drew with rand_like under torch.manual_seed(seed0 + step)."""
    us = []
    for step in range(sc["steps"]):
        torch.manual_seed(seed0 + step)
        us.append(torch.rand(sc["H"], sc["W"], sc["n"]).reshape(-1, sc["n"]).to(dev))
    return us


def _run(sc, us, dev, make_model, make_opt, make_sched, loss_of, clip, mode):
    """mode: 'eager' (scheduler over FusedAdam), 'graph' (the same through GraphedStep), 'frozen' (no scheduler)."""
    from nerf_amd import ops
    from nerf_amd.graph import GraphedStep
    from nerf_amd.render import render_rays
    torch.manual_seed(0)
    m = make_model().to(dev)
    params = list(m.parameters())
    opt = make_opt(params, mode == "graph")
    sched = make_sched(opt) if mode != "frozen" else None
    R = sc["H"] * sc["W"]
    o_s, d_s = torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev)
    gt_s, u_s = torch.empty(R, 3, device=dev), torch.empty(R, sc["n"], device=dev)
    z_lin = torch.linspace(sc["near"], sc["far"], sc["n"]).to(dev)

    def load(step):
        v = step % sc["views"]
        o_s.copy_(sc["rays"][v][0]); d_s.copy_(sc["rays"][v][1]); gt_s.copy_(sc["imgs"][v]); u_s.copy_(us[step])

    def step_fn():
        out = render_rays(m, o_s, d_s, sc["near"], sc["far"], sc["n"], t_rand=u_s, z_lin=z_lin)
        loss = loss_of(out, gt_s)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if clip:
            ops.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss.detach()

    losses, lrs = [], []
    first = 0
    run = step_fn
    if mode == "graph":
        warm = []

        def prepare(i):
            load(i)
            lrs.append(opt.param_groups[0]["lr"])

        # the warm-up steps are steps 0 and 1 of the run (no scheduler step falls between them: epochs are >= 2 views)
        class Collect:
            def __call__(self):
                l = step_fn()
                if not torch.cuda.is_current_stream_capturing():
                    warm.append(l.clone())
                return l
        run = GraphedStep(Collect(), warmup=2, optimizers=[opt], prepare=prepare)
        losses += warm
        first = 2
        for i in range(2):                                   # the scheduler steps those two steps' loop would have taken
            if i % sc["views"] == sc["views"] - 1:
                sched.step()
    for step in range(first, sc["steps"]):
        load(step)
        lrs.append(opt.param_groups[0]["lr"])
        losses.append(run().clone())
        if sched is not None and step % sc["views"] == sc["views"] - 1:
            sched.step()
    return torch.stack(losses).cpu().double().numpy(), np.array(lrs), m


def _epoch_psnr(losses, views):
    return (-10.0 * np.log10(losses)).reshape(-1, views).mean(axis=1)


def test_exponential_lr_over_vanilla_matches_the_reference_trajectory(dev):
    from nerf_amd.models.NeRF import NeRF
    from nerf_amd.optim import FusedAdam
    g = load_golden("g10_sched.npz")
    sc = _scene(g, "a", dev)
    us = _uniforms(sc, 2000, dev)
    want, want_lr = g["a_loss"], g["a_lr"]
    mse = lambda out, gt: torch.nn.functional.mse_loss(out["rgb"], gt)
    res = {}
    for mode in ("eager", "graph", "frozen"):
        res[mode] = _run(sc, us, dev, NeRF, lambda ps, cap: FusedAdam(ps, lr=5e-4, capturable=cap),
                         lambda o: torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.98), mse, True, mode)
    for mode in ("eager", "graph"):
        losses, lrs, _ = res[mode]
        assert np.allclose(lrs, want_lr, rtol=1e-12), mode                          # the rates the reference's loop had
        rel = np.abs(losses[:20] - want[:20]) / want[:20]
        ep, ep_ref = _epoch_psnr(losses, sc["views"]), _epoch_psnr(want, sc["views"])
        print(mode, "max rel loss error over 20 steps %.2e" % rel.max(), "epoch psnr", np.round(ep, 3), "reference", np.round(ep_ref, 3))
        assert rel.max() < 1e-3, (mode, rel)
        assert abs(ep[-1] - ep_ref[-1]) < 0.1, (mode, ep, ep_ref)
    # the test can tell: the same run at a constant rate leaves the reference's trajectory
    frozen = res["frozen"][0]
    d_sched = np.abs(np.log(res["eager"][0][40:] / want[40:])).mean()
    d_frozen = np.abs(np.log(frozen[40:] / want[40:])).mean()
    print("mean |log loss ratio| over steps 40..: scheduled %.2e, constant rate %.2e" % (d_sched, d_frozen))
    assert d_frozen > 3.0 * d_sched


def test_cosine_lr_over_hash_grid_model_matches_the_reference_trajectory(dev):
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.optim import FusedAdam
    from nerf_amd.render import ngp_loss
    g = load_golden("g10_sched.npz")
    sc = _scene(g, "b", dev)
    us = _uniforms(sc, 3000, dev)
    want, want_lr = g["b_loss"], g["b_lr"]
    res = {}
    for mode in ("eager", "graph", "frozen"):
        res[mode] = _run(sc, us, dev, FastNGPNeRF, lambda ps, cap: FusedAdam(ps, lr=1e-2, eps=1e-15, capturable=cap),
                         lambda o: torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=30, eta_min=1e-4), ngp_loss, False, mode)
    for mode in ("eager", "graph"):
        losses, lrs, _ = res[mode]
        assert np.allclose(lrs, want_lr, rtol=1e-12), mode
        rel = np.abs(losses[:20] - want[:20]) / want[:20]
        ep, ep_ref = _epoch_psnr(losses, sc["views"]), _epoch_psnr(want, sc["views"])
        print(mode, "max rel loss error over 20 steps %.2e" % rel.max(), "last epochs", np.round(ep[-3:], 3), "reference", np.round(ep_ref[-3:], 3))
        assert rel.max() < 1e-3, (mode, rel)
        assert abs(ep[-1] - ep_ref[-1]) < 0.1, (mode, ep, ep_ref)
    frozen = res["frozen"][0]
    d_sched = np.abs(np.log(res["eager"][0][20:] / want[20:])).mean()
    d_frozen = np.abs(np.log(frozen[20:] / want[20:])).mean()
    print("mean |log loss ratio| over steps 20..: scheduled %.2e, constant rate %.2e" % (d_sched, d_frozen))
    assert d_frozen > 3.0 * d_sched


def test_replays_follow_the_learning_rate(dev):
    """Two replays of one captured step at different rates move the parameters by different amounts: with a constant
    gradient Adam's update is the rate itself (m / sqrt(v) = 1 once the bias corrections cancel)."""
    from nerf_amd.graph import GraphedStep
    from nerf_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(1000, device=dev))
    p.grad = torch.full_like(p, 0.5)
    opt = FusedAdam([p], lr=1e-2, capturable=True)
    g = GraphedStep(lambda: opt.step(), warmup=1, optimizers=[opt])
    moves = []
    for lr in (1e-2, 1e-3, 4e-2):
        opt.param_groups[0]["lr"] = lr
        before = p.detach().clone()
        g()
        moves.append(float((p.detach() - before).abs().mean()))
    assert np.allclose(moves, [1e-2, 1e-3, 4e-2], rtol=1e-5), moves
    assert float(opt.state[p]["step"]) == 4.0
    # a rate changed behind the optimiser's back must not be captured silently
    opt.param_groups[0]["lr"] = 7e-3
    with pytest.raises(Exception):
        GraphedStepNoSync(opt)
    opt.sync_lr()
    with pytest.raises(ValueError):
        GraphedStep(lambda: None, optimizers=[FusedAdam([p], lr=1e-2)])


def GraphedStepNoSync(opt):
    """Capture opt.step() without the sync in front of the capture (what a hand-written capture would do)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph):
                opt.step()
    finally:
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()


def test_capturable_state_loads_into_per_parameter_step_optimisers(dev):
    """A capturable group's parameters share one step tensor; the state dict carries a copy per parameter, so that
    optimisers advancing each parameter's step by itself keep the right bias correction."""
    from nerf_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(3)
    shapes = [(64, 7), (64,), (5,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(4)).to(dev)) for s in shapes]
    a, b, c, d = mk(), mk(), mk(), mk()
    o_a = FusedAdam(a, lr=1e-2, capturable=True)
    o_ref = torch.optim.Adam(d, lr=1e-2)
    grads = [[(torch.randn(s, generator=gen) * 0.1).to(dev) for s in shapes] for _ in range(5)]
    for it in range(3):
        for ps in (a, d):
            for p, gr in zip(ps, grads[it]):
                p.grad = gr.clone()
        o_a.step(); o_ref.step()
    sd = o_a.state_dict()
    steps = [sd["state"][k]["step"] for k in sd["state"]]
    assert len({s.data_ptr() for s in steps}) == len(steps) and all(float(s) == 3.0 for s in steps)
    o_b = FusedAdam(b, lr=1e-2)                      # steps counted on the host, one per parameter
    o_c = torch.optim.Adam(c, lr=1e-2)
    for ps in (b, c):
        with torch.no_grad():
            for p, q in zip(ps, a):
                p.copy_(q)
    import copy
    o_b.load_state_dict(copy.deepcopy(sd)); o_c.load_state_dict(copy.deepcopy(sd))   # (load_state_dict adopts the tensors it is given)
    for it in range(3, 5):
        for ps in (a, b, c, d):
            for p, gr in zip(ps, grads[it]):
                p.grad = gr.clone()
        o_a.step(); o_b.step(); o_c.step(); o_ref.step()
    assert float(o_b.state[b[0]]["step"]) == 5.0 and float(o_c.state[c[0]]["step"]) == 5.0
    for pa, pb, pc, pd in zip(a, b, c, d):
        for q in (pa, pb, pc):
            assert torch.allclose(q, pd, rtol=3e-6, atol=1e-7), float((q.detach() - pd.detach()).abs().max())


@pytest.mark.parametrize("which", ["step", "plateau", "exponential"])
def test_torch_schedulers_drive_fused_adam_like_torch_adam(dev, which):
    """StepLR(100, 0.5) (src/train_nerf_llff.py:83, src/train_nerf_skow.py:48), ReduceLROnPlateau(mode='max', factor=0.3,
    patience=4, threshold=0.003, threshold_mode='abs', min_lr=3e-4) (src/train_nerf_ngp.py:57-61) and ExponentialLR over
    FusedAdam -- eager and capturable -- against the same scheduler over torch.optim.Adam: the same rates, the same
    parameters."""
    from nerf_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(17)
    shapes = [(64, 33), (64,), (4099, 2)]
    init = [torch.randn(s, generator=gen) for s in shapes]
    mk = lambda: [torch.nn.Parameter(t.clone().to(dev)) for t in init]

    def sched_of(opt):
        if which == "step":
            return torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
        if which == "plateau":
            return torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.3, patience=1, threshold=0.003,
                                                              threshold_mode="abs", min_lr=3e-4)
        return torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)

    sets = [mk(), mk(), mk()]
    opts = [torch.optim.Adam(sets[0], lr=1e-2, eps=1e-15), FusedAdam(sets[1], lr=1e-2, eps=1e-15),
            FusedAdam(sets[2], lr=1e-2, eps=1e-15, capturable=True)]
    scheds = [sched_of(o) for o in opts]
    metric = [0.30, 0.31, 0.311, 0.3115, 0.3116, 0.40, 0.401, 0.4011, 0.4012, 0.4012, 0.4012, 0.4012]   # a "validation SSIM"
    for it in range(12):
        grads = [(torch.randn(s, generator=gen) * 0.1).to(dev) for s in shapes]
        for ps, o in zip(sets, opts):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
            o.step()
        for sc in scheds:
            sc.step(metric[it]) if which == "plateau" else sc.step()
        lrs = [o.param_groups[0]["lr"] for o in opts]
        assert lrs[0] == lrs[1] == lrs[2], (it, lrs)
        for a, b, c in zip(*sets):
            assert torch.allclose(a, b, rtol=3e-6, atol=1e-7) and torch.allclose(a, c, rtol=3e-6, atol=1e-7), it
    assert opts[0].param_groups[0]["lr"] < 1e-2      # the schedule did something
