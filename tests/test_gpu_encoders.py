"""HIP encoders vs the CPU oracle and the reference's golden vectors."""
import pytest
import torch

from conftest import T, load_golden, rel_err
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def test_pe_golden_and_worst_case_arguments(dev):
    from nerf_amd.utils.positional_encoding import PositionalEncoding
    g = load_golden("g3_pe.npz")
    x = T(g["x"]).to(dev)
    pe10, pe4 = PositionalEncoding(10), PositionalEncoding(4)
    assert pe10.output_dims == 63 and pe4.output_dims == 27
    # tolerance 1e-6: arguments reach 2^9 * 6 = 3072 rad, so this also checks the range reduction
    assert float((pe10(x).cpu() - T(g["pe10"])).abs().max()) < 1e-6
    assert float((pe4(x).cpu() - T(g["pe4"])).abs().max()) < 1e-6
    assert "freq_bands" not in pe10.state_dict()
    gen = torch.Generator().manual_seed(3)
    xb = (torch.rand(5000, 7, 3, generator=gen) * 12 - 6)
    out = pe10(xb.to(dev))
    assert out.shape == (5000, 7, 63)
    assert float((out.cpu() - O.positional_encoding(xb, 10)).abs().max()) < 1e-6


def _filled_tables(sizes, dev):
    tabs = []
    for l, n in enumerate(sizes):
        r = torch.arange(int(n), dtype=torch.float64)
        tabs.append(torch.stack([(torch.sin(r * (0.37 + 0.011 * l) + 1.3 * f + 0.5 * l) * 0.5).float()
                                 for f in range(2)], dim=1).to(dev))
    return tabs


def test_hashgrid_indices_bit_exact_vs_golden(dev):
    from nerf_amd import ops
    g = load_golden("g4_hash.npz")
    res = [int(v) for v in g["resolutions"]]
    sizes = [int(v) for v in g["sizes"]]
    tabs = _filled_tables(sizes, dev)
    x = T(g["x"]).to(dev)
    feats, idx, w = ops.hashgrid_debug(x, tabs, res)
    assert torch.equal(idx.cpu(), T(g["idx"]))        # hash-table indices: bit-exact
    assert torch.equal(w.cpu(), T(g["w"]))            # interpolation weights: bit-exact
    assert float((feats.cpu() - T(g["feats"])).abs().max()) < 1e-6
    # backward into the tables
    tg = [t.clone().requires_grad_(True) for t in tabs]
    out = ops.hashgrid_encode(x, tg, res)
    assert torch.equal(out.detach(), feats)
    (out * T(g["a"]).to(dev)).sum().backward()
    for l, t in enumerate(tg):
        # tolerance: atomics reorder float sums; 1e-3 relative is the contract
        assert abs(float(t.grad.double().norm()) - g["grad_l2"][l]) <= 1e-5 * g["grad_l2"][l]
        assert abs(float(t.grad.double().sum()) - g["grad_sum"][l]) <= 1e-4 * g["grad_l2"][l]
        rows = T(g["grad_rows"][l]).long().to(dev)
        assert torch.allclose(t.grad[rows].cpu(), T(g["grad_vals"][l]), rtol=1e-4, atol=1e-6)


def test_hashgrid_module_state_dict_and_oracle(dev):
    from nerf_amd.utils.hash_grid_encoder import HashGridEncoder
    torch.manual_seed(0)
    ref = O.HashGridOracle()
    torch.manual_seed(0)
    enc = HashGridEncoder()
    sd, sd_ref = enc.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(sd_ref.keys())
    assert all(torch.equal(sd[k], sd_ref[k]) for k in sd)
    assert sd["primes"].dtype == torch.int64
    enc = enc.to(dev)
    gen = torch.Generator().manual_seed(8)
    x = torch.rand(20000, 3, generator=gen) * 2.4 - 1.2   # includes negative and >1 coordinates
    out = enc(x.to(dev))
    out_ref = ref(x)
    assert float((out.cpu() - out_ref).abs().max()) < 1e-8   # table entries are <= 1e-3
    a = torch.randn(20000, 32, generator=gen)
    (out * a.to(dev)).sum().backward()
    (out_ref * a).sum().backward()
    for e, r in zip(enc.embeddings, ref.embeddings):
        assert rel_err(e.weight.grad.cpu(), r.weight.grad) < 1e-5
    assert torch.equal(enc.hash_fn(torch.tensor([[3, 5, 7]], device=dev), 4096).cpu(),
                       ref.primes.mul(torch.tensor([[3, 5, 7]])).sum(-1).bitwise_and(0xFFFFFFFF) % 4096)


def test_hashgrid_empty_input(dev):
    from nerf_amd import ops
    tabs = _filled_tables([64, 64], dev)
    out = ops.hashgrid_encode(torch.zeros(0, 3, device=dev), tabs, [4, 4])
    assert out.shape == (0, 4)


def _scatter(dev, x, d_out, res, sizes, binned):
    """nerf_hashgrid_bwd through the C ABI, with or without the binning workspace."""
    from nerf_amd import _lib as L, ops
    grads = [torch.zeros(s, 2, device=dev) for s in sizes]
    work, nbytes = ops.scatter_workspace(x.shape[0], sizes, dev) if binned else (None, 0)
    assert not binned or nbytes > 0
    L.check(L.lib().nerf_hashgrid_bwd(L.ptr(x), x.shape[0], L.ptr(d_out), L.int_array(res), L.int_array(sizes), len(res),
                                      L.ptr_array(grads), L.ptr(work), nbytes, L.stream_of(x)), "nerf_hashgrid_bwd")
    return grads


def test_binned_scatter_matches_direct_atomics_and_oracle(dev):
    """The table gradient with its updates sorted by table segment and summed in LDS: the same sums as one float
    atomic per update, and as the oracle's dense Embedding gradient (points outside [0,1] and negative ones included:
    wrapped cells put the two x corners of a pair into rows that are not neighbours)."""
    gen = torch.Generator().manual_seed(21)
    res = O.hash_resolutions()
    sizes = O.hash_table_sizes(res)
    P = 70001                                               # above the binning threshold, not a multiple of anything
    x = torch.rand(P, 3, generator=gen) * 2.4 - 1.2
    x[:64] = torch.tensor([0.999999, -1.0, 0.5])            # many points in one cell: one segment gets a run of equal rows
    # coordinates just below zero: x * res - floor(x * res) rounds to exactly 1, the low-x weight to 0 and their ratio
    # to infinity -- such corners must take the direct path, or 0 * inf would poison the table gradient
    x[64:128, 0] = -1e-9
    x[128:192] = torch.tensor([-1e-10, -3e-9, -1e-12])
    x[192:256, 0] = -(2.0 ** -26) / torch.tensor(res, dtype=torch.float32).repeat(4)
    d_out = torch.randn(P, 32, generator=gen)
    xd, dd = x.to(dev), d_out.to(dev)
    direct = _scatter(dev, xd, dd, res, sizes, binned=False)
    binned = _scatter(dev, xd, dd, res, sizes, binned=True)
    for l, (a, b) in enumerate(zip(binned, direct)):
        assert bool(torch.isfinite(a).all()), l
        assert rel_err(a, b) < 2e-6, l
    tabs = [torch.zeros(s, 2, requires_grad=True) for s in sizes]
    (O.hash_encode(x, tabs, res) * d_out).sum().backward()
    for l, (a, t) in enumerate(zip(binned, tabs)):
        assert rel_err(a.cpu(), t.grad) < 1e-5, l


def test_binned_scatter_survives_a_full_queue(dev):
    """All points in one cell: every record of a level lands in ONE segment, far beyond that queue's capacity
    (1.25 x the even share); the overflow is applied directly.  Also a table-size mix with a last partial segment."""
    res = [40, 101, 512]
    sizes = [64000, 524288, 20000]                          # 4 segments (last partial), 32 segments, 2 segments
    P = 66000
    gen = torch.Generator().manual_seed(22)
    x = torch.full((P, 3), 0.3137) + torch.rand(P, 3, generator=gen) * 1e-4
    d_out = torch.randn(P, 6, generator=gen)
    xd, dd = x.to(dev), d_out.to(dev)
    direct = _scatter(dev, xd, dd, res, sizes, binned=False)
    binned = _scatter(dev, xd, dd, res, sizes, binned=True)
    for l, (a, b) in enumerate(zip(binned, direct)):
        assert int((b != 0).any(dim=1).sum()) <= 8          # one cell: at most 8 rows
        # a handful of rows each summing 66,000 terms in another order
        assert rel_err(a, b) < 5e-5, l
    tabs = [torch.zeros(s, 2, dtype=torch.float64, requires_grad=True) for s in sizes]
    (O.hash_encode(x.double(), tabs, res) * d_out.double()).sum().backward()
    for l, (a, t) in enumerate(zip(binned, tabs)):
        assert rel_err(a.cpu(), t.grad) < 5e-5, l


def test_scatter_workspace_is_checked(dev):
    """A workspace that is too small or not 16-byte aligned is refused with a message before anything is launched."""
    from nerf_amd import _lib as L, ops
    res, sizes = O.hash_resolutions(), O.hash_table_sizes(O.hash_resolutions())
    P = 70000
    x = torch.rand(P, 3, device=dev)
    d_out = torch.randn(P, 32, device=dev)
    grads = [torch.zeros(s, 2, device=dev) for s in sizes]
    work, nbytes = ops.scatter_workspace(P, sizes, dev)
    args = lambda w, n: (L.ptr(x), P, L.ptr(d_out), L.int_array(res), L.int_array(sizes), 16, L.ptr_array(grads), w, n,
                         L.stream_of(x))
    with pytest.raises(L.NerfHipError, match="needed"):
        L.check(L.lib().nerf_hashgrid_bwd(*args(L.ptr(work), nbytes // 2)), "nerf_hashgrid_bwd")
    with pytest.raises(L.NerfHipError, match="aligned"):
        L.check(L.lib().nerf_hashgrid_bwd(*args(work.data_ptr() + 4, nbytes)), "nerf_hashgrid_bwd")
    torch.cuda.synchronize()
    assert all(float(g.abs().max()) == 0.0 for g in grads)          # nothing was launched
    L.check(L.lib().nerf_hashgrid_bwd(*args(L.ptr(work), nbytes)), "nerf_hashgrid_bwd")
    assert float(grads[-1].abs().max()) > 0.0
