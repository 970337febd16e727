"""A training step captured into a HIP graph (nerf_amd.graph.GraphedStep) against the same steps run eagerly."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_capturable_adam_matches_torch_adam(dev):
    from nerf_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(13)
    shapes = [(256, 63), (256,), (4099, 2), (0, 5), (3,)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=gen).to(dev)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-2, eps=1e-15, weight_decay=1e-6)
    o_mine = FusedAdam(mine, lr=1e-2, eps=1e-15, weight_decay=1e-6, capturable=True)
    for it in range(6):
        for a, b in zip(ref, mine):
            g = (torch.randn(a.shape, generator=gen) * 0.1).to(dev)
            a.grad, b.grad = g, g.clone()
        o_ref.step()
        o_mine.step()
        for a, b in zip(ref, mine):
            # the two step-dependent scalars come from the device's double-precision pow here
            assert torch.allclose(a, b, rtol=3e-6, atol=1e-7), (it, tuple(a.shape), float((a - b).abs().max()))
    st = o_mine.state[mine[0]]["step"]
    assert st.is_cuda and float(st) == 6.0
    sd = o_mine.state_dict()
    o2 = FusedAdam(mine, lr=1e-2, eps=1e-15, weight_decay=1e-6, capturable=True)
    o2.load_state_dict(sd)
    for b in mine:
        b.grad = torch.zeros_like(b)
    o2.step()
    assert float(o2.state[mine[0]]["step"]) == 7.0


def _ngp_setup(dev, rays=1024, capturable=False):
    from nerf_amd.dist import GradBuckets
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    from nerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    model = FastNGPNeRF().to(dev)
    buckets = GradBuckets([list(model.hash_encoder.parameters()),
                           [p for n, p in model.named_parameters() if not n.startswith("hash_encoder")]])
    opt = FusedAdam(buckets.parameters(), lr=1e-2, eps=1e-15, capturable=capturable)
    gen = torch.Generator().manual_seed(5)
    o = (torch.rand(rays, 3, generator=gen) * 0.2 - 0.1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=gen), dim=-1).to(dev)
    u = torch.rand(rays, 32, generator=gen).to(dev)
    gt = torch.rand(rays, 3, generator=gen).to(dev)
    z_lin = torch.linspace(0.05, 1.2, 32).to(dev)
    return model, buckets, opt, o, d, u, gt, z_lin


def test_graphed_hash_grid_step_matches_eager_steps(dev):
    """Six steps of the hash-grid model (sampling, gather, MLPs, compositing, backward with the binned
    table-gradient scatter, Adam): two eager steps + four replays of a captured step against six eager ones.
    The gradients are sums of float atomics whose order varies from run to run, and Adam with eps = 1e-15 turns a
    gradient that cancels to rounding noise into a full step of either sign -- so two EAGER runs already differ in
    a few entries by multiples of the learning rate.  The eager run is therefore done twice and the captured
    run must be as close to it as the second eager run is (a replay that dropped or doubled a step would be off
    by the size of the parameters themselves)."""
    from nerf_amd.graph import GraphedStep
    from nerf_amd.render import train_step

    def make(capturable):
        model, buckets, opt, o, d, u, gt, z_lin = _ngp_setup(dev, capturable=capturable)

        def step():
            loss = train_step(model, o, d, gt, 0.05, 1.2, 32, buckets=buckets, t_rand=u, z_lin=z_lin, white_bkgd=False)
            opt.step()
            return loss
        return model, opt, step

    def rel_l2(a, b):
        return float((a.detach() - b.detach()).norm() / (a.detach().norm() + 1e-30))

    m_a, _, step_a = make(False)
    losses_a = [float(step_a()) for _ in range(6)]
    m_b, _, step_b = make(False)
    for _ in range(6):
        step_b()
    m_g, opt_g, step_g = make(True)
    g = GraphedStep(step_g, warmup=2)           # 2 eager steps; the capture itself does not execute
    losses_g = [float(g()) for _ in range(4)]   # steps 3..6
    assert float(opt_g.state[next(iter(m_g.parameters()))]["step"]) == 6.0
    assert losses_a[5] < losses_a[0]            # it trains
    for a, b in zip(losses_a[2:], losses_g):
        assert abs(a - b) < 1e-4 * max(1.0, abs(a)), (losses_a, losses_g)
    for (n, a), b, c in zip(m_a.named_parameters(), m_b.parameters(), m_g.parameters()):
        assert rel_l2(a, c) <= max(5.0 * rel_l2(a, b), 2e-3), (n, rel_l2(a, c), rel_l2(a, b))


def test_replays_are_unaffected_by_kernels_and_allocations_between_them(dev):
    """A training loop copies new data in, draws numbers and logs between replays.  131,072 points: the size at which
    the table gradients go through the binned scatter, whose queue counters must be cleared by every replay (a
    captured ``hipMemsetAsync`` stopped doing so once other kernels had run between two replays: queues looked full,
    the reduce pass added the previous replay's records again).  With a learning rate of zero every replay computes
    the same gradients; float atomics leave their order open, hence 1e-5."""
    import gc

    from nerf_amd.graph import GraphedStep
    from nerf_amd.render import train_step
    def build():   # the step's inputs are locals of this call: only the GraphedStep keeps them (and the step function) alive
        model, buckets, opt, o, d, u, gt, z_lin = _ngp_setup(dev, rays=4096, capturable=True)
        for group in opt.param_groups:
            group["lr"] = 0.0

        def step():
            loss = train_step(model, o, d, gt, 0.05, 1.2, 32, buckets=buckets, t_rand=u, z_lin=z_lin, white_bkgd=False)
            opt.step()
            return loss

        loss_eager = float(step())   # the reference: the same step issued launch by launch
        grads_eager = [f.clone() for f in buckets.flats]
        return GraphedStep(step, warmup=2, optimizers=[opt]), buckets, loss_eager, grads_eager

    g, buckets, loss_ref, ref = build()
    gc.collect()
    assert all(float(r.abs().max()) > 0 for r in ref)
    for _ in range(40):
        g()
    for it in range(3):
        torch.cuda.synchronize()   # (as a loop that times or logs does; the stale state needed one)
        junk = [torch.full((1 << 18,), 7.0 + it, device=dev) for _ in range(8)]     # new blocks, written
        noise = torch.rand(4096, 64, device=dev)
        copy = noise.clone().mul_(2.0)
        del junk
        for again in range(3):   # (the stale state showed from the second replay on)
            loss = float(g())
            assert abs(loss - loss_ref) <= 1e-6 * max(1.0, abs(loss_ref)), (it, again, loss, loss_ref)
            for f, r in zip(buckets.flats, ref):
                assert float((f - r).norm()) <= 1e-5 * float(r.norm()), (it, again, float((f - r).norm() / r.norm()))
        del noise, copy
