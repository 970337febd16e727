"""Fused hash-grid NeRF (HIP) vs the reference's golden vectors and the CPU oracle."""
import pytest
import torch

from conftest import T, load_golden, rel_err
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def _grad_close(got, ref, P, name, tol=1e-4):
    """Contract: 1e-3 relative.  Float atomics reorder the sums; a cancelling sum
    over points is compared against its float32 noise floor instead."""
    err = float((got.double() - ref.double()).norm())
    scale = float(ref.double().norm())
    assert err <= tol * scale or err <= 2e-7 * P * got.numel() ** 0.5 * max(1.0, float(ref.abs().max())), (name, err, scale)


def _check_vs_golden(dev, g, name, model, x, d):
    rgb, sigma = model(x.to(dev), d.to(dev))
    assert rgb.shape == g[f"{name}_rgb"].shape and sigma.shape == g[f"{name}_sigma"].shape
    assert float((rgb.detach().cpu() - T(g[f"{name}_rgb"])).abs().max()) < 2e-6
    assert rel_err(sigma.detach().cpu(), T(g[f"{name}_sigma"])) < 2e-6
    loss = (rgb * T(g[f"{name}_a"]).to(dev)).sum() + (sigma * T(g[f"{name}_b"]).to(dev)).sum()
    model.zero_grad()
    loss.backward()
    for k, p in model.named_parameters():
        gr = p.grad.reshape(-1).cpu()
        ref_n = float(g[f"{name}_gnorm/{k}"][0])
        assert abs(float(gr.double().norm()) - ref_n) <= 1e-4 * ref_n + 1e-12, (k, float(gr.double().norm()), ref_n)
        ii = T(g[f"{name}_gidx/{k}"]).long()
        assert torch.allclose(gr[ii], T(g[f"{name}_gval/{k}"]), rtol=1e-3,
                              atol=1e-4 * ref_n / max(1.0, gr.numel() ** 0.25) + 1e-12), k


def test_ngp_vs_reference_golden(dev):
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    g = load_golden("g5_ngp.npz")
    x, d = T(g["x"]), T(g["d"])
    torch.manual_seed(0)
    m = FastNGPNeRF()
    sd = m.state_dict()
    for k, v in sd.items():  # seeded construction reproduces the reference's 10.4M parameters
        ck = g[f"ngp_ck/{k}"]
        assert float(v.double().sum()) == ck[0] and float(v.double().abs().sum()) == ck[1], k
    assert sum(p.numel() for p in m.parameters()) == 10442933 and sd["hash_encoder.primes"].dtype == torch.int64
    m = m.to(dev)
    _check_vs_golden(dev, g, "ngp", m, x, d)
    assert rel_err(m.query_density(x.to(dev)).cpu(), T(g["ngp_density"])) < 2e-6
    with torch.no_grad():
        rgb0, _ = m(x[:16].to(dev), None)
    assert float((rgb0.cpu() - T(g["ngp_nodir_rgb"])).abs().max()) < 1e-6
    # the reference's init makes the outputs nearly constant; scale the weights up for a sharper check
    with torch.no_grad():
        for k, p in m.named_parameters():
            for pat, s in zip(g["boost_keys"], g["boost_vals"]):
                if k.startswith(str(pat)):
                    p.mul_(float(s))
    _check_vs_golden(dev, g, "ngpb", m, x, d)


@pytest.mark.parametrize("P", [1, 33, 5000])
def test_ngp_vs_oracle_all_gradients(dev, P):
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    torch.manual_seed(7)
    ref = O.FastNGPNeRFOracle(log2_hashmap_size=14)   # small tables keep the CPU side quick
    m = FastNGPNeRF(log2_hashmap_size=14)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.startswith("hash_encoder"):
                p.mul_(300.0)
            elif k != "fc_density.2.bias":
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() > 1 else 0.1))
    m.load_state_dict(ref.state_dict())
    m = m.to(dev)
    gen = torch.Generator().manual_seed(P)
    x = torch.rand(P, 3, generator=gen) * 2.6 - 1.3       # includes negative and out-of-cube coordinates
    d = torch.randn(P, 3, generator=gen) * 2.0
    a = torch.randn(P, 3, generator=gen)
    b = torch.randn(P, generator=gen)
    rgb_o, sig_o = ref(x, d)
    ((rgb_o * a).sum() + (sig_o * b).sum()).backward()
    rgb, sig = m(x.to(dev), d.to(dev))
    ((rgb * a.to(dev)).sum() + (sig * b.to(dev)).sum()).backward()
    assert sig.shape == (P,)
    assert float((rgb.detach().cpu() - rgb_o.detach()).abs().max()) < 1e-5
    assert rel_err(sig.detach().cpu(), sig_o.detach()) < 1e-5
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert p.grad.shape == q.grad.shape, k
        _grad_close(p.grad.cpu(), q.grad, P, k)


def test_ngp_state_dict_and_modes(dev):
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    torch.manual_seed(0)
    ref = O.FastNGPNeRFOracle(log2_hashmap_size=12)
    torch.manual_seed(0)
    m = FastNGPNeRF(log2_hashmap_size=12)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    assert all(torch.equal(v, ref.state_dict()[k]) for k, v in m.state_dict().items())
    m = m.to(dev)
    x = (torch.rand(300, 3) * 2 - 1).to(dev)
    d = torch.randn(300, 3).to(dev)
    rgb_t, sig_t = m(x, d)
    with torch.no_grad():
        rgb_i, sig_i = m(x, d)
    assert torch.equal(rgb_t.detach(), rgb_i) and torch.equal(sig_t.detach(), sig_i)
    # one direction per ray (dirs_div) equals the expanded per-point directions
    d_ray = d[::10].contiguous()
    rgb_r, _ = m(x, d_ray, dirs_div=10)
    rgb_x, _ = m(x, d_ray[:, None, :].expand(30, 10, 3).reshape(-1, 3))
    assert torch.equal(rgb_r.detach(), rgb_x.detach())
    assert m(x[:0], d[:0])[0].shape == (0, 3)
