"""The one-launch Adam against torch.optim.Adam (the optimiser the reference's loops use)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lr,eps", [(5e-4, 1e-8), (1e-2, 1e-15)])   # src/train_nerf_llff.py:82, src/train_nerf_ngp.py:55
def test_fused_adam_matches_torch_adam(dev, lr, eps):
    from nerf_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(11)
    shapes = [(256, 63), (256,), (1, 256), (3,), (4099, 2), (524288, 2), (0, 5), (7, 13, 3)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=gen).to(dev)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref, o_mine = torch.optim.Adam(ref, lr=lr, eps=eps), FusedAdam(mine, lr=lr, eps=eps)
    for it in range(5):
        for a, b in zip(ref, mine):
            g = (torch.randn(a.shape, generator=gen) * (10.0 ** (it - 3))).to(dev)
            if it == 2 and a.numel() > 100:
                g.view(-1)[::3] = 0.0           # untouched hash-table rows have zero gradient
            a.grad, b.grad = g, g.clone()
        if it == 3:
            ref[1].grad = None                  # a parameter without a gradient is skipped
            mine[1].grad = None
        o_ref.step()
        o_mine.step()
        for a, b in zip(ref, mine):
            # same formula, float32; sqrt/divide may differ in the last bit
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, tuple(a.shape), float((a - b).abs().max()))
    sd = o_mine.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    o_ref.load_state_dict(sd)                   # torch accepts the state as its own
    assert float(o_ref.state[ref[0]]["step"]) == 5.0


def test_fused_adam_rejects_unsupported_options(dev):
    from nerf_amd.optim import FusedAdam
    p = [torch.nn.Parameter(torch.zeros(4, device=dev))]
    with pytest.raises(NotImplementedError):
        FusedAdam(p, amsgrad=True)
    with pytest.raises(ValueError):
        FusedAdam(p, weight_decay=-1.0)


def test_fused_adam_weight_decay_matches_torch(dev):
    """Adam(lr=5e-4, weight_decay=1e-6) of src/train_nerf_llff.py:82 / src/train_nerf_skow.py:47 (and a large decay,
    where a wrong formula would show)."""
    from nerf_amd.optim import FusedAdam
    for wd in (1e-6, 0.3):
        gen = torch.Generator().manual_seed(12)
        shapes = [(256, 319), (256,), (4099, 2), (5,)]
        ref = [torch.nn.Parameter(torch.randn(s, generator=gen).to(dev)) for s in shapes]
        mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
        o_ref = torch.optim.Adam(ref, lr=5e-4, weight_decay=wd)
        o_mine = FusedAdam(mine, lr=5e-4, weight_decay=wd)
        for it in range(4):
            for a, b in zip(ref, mine):
                g = (torch.randn(a.shape, generator=gen) * 0.1).to(dev)
                a.grad, b.grad = g, g.clone()
            o_ref.step()
            o_mine.step()
            for a, b in zip(ref, mine):
                assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (wd, it, tuple(a.shape), float((a - b).abs().max()))
                assert torch.equal(b.grad, a.grad)          # the decayed gradient is not written back


def test_fused_adam_more_tensors_than_one_launch_holds(dev):
    """More than 64 tensors, empty ones among the first 64: every tensor is updated exactly once."""
    from nerf_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(13)
    shapes = [(0,) if k in (3, 40) else (17 + k,) for k in range(150)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=gen).to(dev)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref, o_mine = torch.optim.Adam(ref, lr=1e-2), FusedAdam(mine, lr=1e-2)
    for it in range(3):
        for a, b in zip(ref, mine):
            g = torch.randn(a.shape, generator=gen).to(dev)
            a.grad, b.grad = g, g.clone()
        o_ref.step()
        o_mine.step()
    for k, (a, b) in enumerate(zip(ref, mine)):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), k


@pytest.mark.parametrize("scale", [1e-3, 1.0, 50.0])   # below, near and above max_norm = 1 (src/train_nerf.py:165)
def test_clip_grad_norm_matches_torch(dev, scale):
    from nerf_amd.ops import clip_grad_norm_
    gen = torch.Generator().manual_seed(14)
    shapes = [(256, 63), (256,), (1, 256), (3,), (4099, 2), (524288, 2), (7, 13, 3)] + [(33,)] * 70
    ref = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
    mine = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
    for a, b in zip(ref, mine):
        g = (torch.randn(a.shape, generator=gen) * scale / 1000.0).to(dev)
        a.grad, b.grad = g, g.clone()
    ref[2].grad = None
    mine[2].grad = None
    t_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm=1.0)
    t_mine = clip_grad_norm_(mine, max_norm=1.0)
    assert t_mine.is_cuda and t_mine.dim() == 0
    assert abs(float(t_mine) - float(t_ref)) <= 2e-6 * float(t_ref)
    for a, b in zip(ref, mine):
        if a.grad is None:
            assert b.grad is None
        else:
            assert torch.allclose(a.grad, b.grad, rtol=3e-6, atol=0.0), tuple(a.shape)
    again = clip_grad_norm_(mine, max_norm=1.0)              # deterministic reduction order: same bits every run
    assert float(again) <= 1.0 + 1e-5
