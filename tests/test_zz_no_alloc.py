"""nerf_clip_grad_norm_static (the entry point that allocates nothing: its scratch is a device array built
into the library) against nerf_clip_grad_norm with a caller workspace, through the C ABI."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _call_both(dev, shapes, seed, scale, max_norm=1.0):
    from nerf_amd import _lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(seed)
    a = [(torch.randn(s, generator=gen) * scale).to(dev) for s in shapes]
    b = [t.clone() for t in a]
    n = len(shapes)
    numel = (C.c_int64 * n)(*[t.numel() for t in a])
    work = torch.empty(lib.nerf_clip_grad_norm_workspace_floats(n, numel), device=dev)
    tot_a, tot_b = torch.zeros((), device=dev), torch.zeros((), device=dev)
    st = L.stream_of(a[0])
    L.check(lib.nerf_clip_grad_norm(n, L.ptr_array(a), numel, max_norm, L.ptr(work), L.ptr(tot_a), st), "nerf_clip_grad_norm")
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
    L.check(lib.nerf_clip_grad_norm_static(n, L.ptr_array(b), numel, max_norm, L.ptr(tot_b), st), "nerf_clip_grad_norm_static")
    torch.cuda.synchronize()
    assert (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)) == before
    assert float(tot_a) == float(tot_b) and float(tot_a) > 0.0
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    return float(tot_a)


def test_static_scratch_entry_point_matches_workspace_entry_point(dev):
    big = [(256, 63), (256,), (524288, 2), (4099, 2), (1, 256), (3,)] + [(33,)] * 70     # 333 chunks, two launches per pass
    small = [(128, 283), (128,), (5,)]                                                    # 11 chunks: the scratch keeps stale partials
    t1 = _call_both(dev, big, seed=21, scale=0.05)      # norm well above 1: gradients are scaled
    t2 = _call_both(dev, small, seed=22, scale=0.05)
    t3 = _call_both(dev, small, seed=23, scale=1e-4)    # norm below 1: gradients untouched
    assert t1 > 1.0 and t2 > 1.0 and t3 < 1.0
    # against torch, on the second (smaller) call's data
    gen = torch.Generator().manual_seed(22)
    g = [(torch.randn(s, generator=gen) * 0.05) for s in small]
    want = float(torch.linalg.vector_norm(torch.cat([t.reshape(-1) for t in g]).double()))
    assert abs(t2 - want) <= 2e-6 * want


def test_static_scratch_entry_point_rejects_oversized_lists(dev):
    from nerf_amd import _lib as L
    lib = L.lib()
    t = torch.zeros(16, device=dev)
    numel = (C.c_int64 * 1)((65536 + 1) * 4096)          # one chunk more than the built-in scratch holds; never dereferenced
    rc = lib.nerf_clip_grad_norm_static(1, L.ptr_array([t]), numel, 1.0, None, L.stream_of(t))
    assert rc == -22
    assert b"exceed the built-in scratch" in lib.nerf_last_error()
    torch.cuda.synchronize()
    assert float(t.abs().sum()) == 0.0
    numel_ok = (C.c_int64 * 1)(16)
    assert lib.nerf_clip_grad_norm_static(1, L.ptr_array([t]), numel_ok, 1.0, None, L.stream_of(t)) == 0
    torch.cuda.synchronize()
