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
        FusedAdam(p, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        FusedAdam(p, amsgrad=True)
