"""The gradient exchange of the data-parallel step on real hardware: one rank, backend nccl (RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bucketed_chunked_step_under_nccl_matches_plain_step(dev):
    """The all-reduces of the flat gradient buckets are issued from gradient hooks right behind our backward
    kernels.  With one rank the sum is the identity, so the bucketed, chunked, hook-driven step under RCCL must
    reproduce the gradients of the plain single-process step -- which it only does if the collectives are
    ordered correctly against the kernels on the stream (and it must not hang)."""
    from nerf_amd import render
    from nerf_amd.dist import GradBuckets
    from nerf_amd.models.NeRF import NeRF
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(3)
        coarse, fine = NeRF().to(dev), NeRF().to(dev)
        R = 1024
        g = torch.Generator().manual_seed(4)
        o = torch.tensor([0.0, 0.0, 4.0]).expand(R, 3).contiguous().to(dev)
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * torch.tensor([0.3, 0.3, 0.0]) +
                                          torch.tensor([0.0, 0.0, -1.0]), dim=-1).to(dev)
        tgt = torch.rand(R, 3, generator=g).to(dev)
        t_rand = torch.rand(R, 64, generator=g).to(dev)
        u_fine = torch.rand(R, 128, generator=g).to(dev)
        kw = dict(fine_model=fine, n_importance=128, t_rand=t_rand, u_fine=u_fine)
        loss_ref = render.train_step(coarse, o, d, tgt, 2.0, 6.0, 64, **kw)
        ref = [p.grad.clone() for m in (fine, coarse) for p in m.parameters()]
        gb = GradBuckets([list(fine.parameters()), list(coarse.parameters())], always_reduce=True)
        for _ in range(2):   # the second pass reuses the buckets
            loss = render.train_step(coarse, o, d, tgt, 2.0, 6.0, 64, buckets=gb, chunk_rays=256, **kw)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(loss_ref)) < 1e-6
        for p, r in zip(gb.parameters(), ref):
            err = float((p.grad - r).double().norm())
            # 256 = 2^8 rays per chunk: the chunk weights are exact; the sums differ by their order only
            assert err <= 2e-5 * float(r.double().norm()) + 1e-12, err
    finally:
        dist.destroy_process_group()


def test_hash_grid_step_sends_the_tables_before_the_weight_gradients(dev):
    """The hash-grid model's backward reports its table gradients after the scatter and its MLP gradients after the
    weight-gradient kernels; with buckets [tables], [MLP parameters] the tables' all-reduce is therefore issued
    between the two phases (and runs under the second).  One rank under RCCL: the sums are identities, so the
    gradients must equal those of the plain step -- which holds only if each collective is ordered behind the
    kernels that fill its bucket -- and the order of the sends is recorded."""
    from nerf_amd import render
    from nerf_amd.dist import GradBuckets
    from nerf_amd.models.FastNGPNeRF import FastNGPNeRF
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(5)
        model = FastNGPNeRF().to(dev)
        R = 2048
        g = torch.Generator().manual_seed(6)
        o = (torch.rand(R, 3, generator=g) * 0.2 - 0.1).to(dev)
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
        tgt = torch.rand(R, 3, generator=g).to(dev)
        t_rand = torch.rand(R, 64, generator=g).to(dev)
        kw = dict(t_rand=t_rand, loss_fn=render.ngp_loss)
        for p in model.parameters():
            p.grad = None
        loss_ref = render.train_step(model, o, d, tgt, 0.05, 1.2, 64, **kw)
        ref = {n: p.grad.clone() for n, p in model.named_parameters()}
        tables = list(model.hash_encoder.parameters())
        rest = [p for n, p in model.named_parameters() if not n.startswith("hash_encoder")]
        gb = GradBuckets([tables, rest], always_reduce=True)
        order = []
        send = gb._send
        gb._send = lambda bi: (order.append((bi, list(gb._pending))), send(bi))[1]
        for _ in range(2):
            order.clear()
            loss = render.train_step(model, o, d, tgt, 0.05, 1.2, 64, buckets=gb, **kw)
        torch.cuda.synchronize()
        assert [bi for bi, _ in order] == [0, 1]            # tables first
        assert order[0][1][1] > 0                           # ... while the MLP parameters' deliveries were still to come
        assert abs(float(loss) - float(loss_ref)) < 1e-6
        for n, p in model.named_parameters():
            r = ref[n]
            err = float((p.grad - r).double().norm())
            assert err <= 2e-5 * float(r.double().norm()) + 1e-12, (n, err)
    finally:
        dist.destroy_process_group()


def test_c_abi_allreduce_with_one_rank(dev):
    """nerf_allreduce_sum over a communicator made by nerf_comm_unique_id / nerf_comm_init_rank (world 1: the sum is
    the identity), enqueued behind a kernel of ours on the same stream; a second call reuses the communicator."""
    import ctypes as C
    from nerf_amd import _lib as L
    lib = L.lib()
    assert lib.nerf_rccl_version() >= 20000
    ident = (C.c_char * 128)()
    L.check(lib.nerf_comm_unique_id(ident), "nerf_comm_unique_id")
    comm = C.c_void_p()
    torch.cuda.set_device(dev)
    L.check(lib.nerf_comm_init_rank(ident, 1, 0, C.byref(comm)), "nerf_comm_init_rank")
    try:
        assert comm.value
        gen = torch.Generator().manual_seed(8)
        for n in (1, 1000, 10_435_874 + 7059):           # up to the hash-grid model's whole gradient
            buf = torch.randn(n, generator=gen).to(dev)
            want = buf * 2.0
            buf.mul_(2.0)                                 # a kernel on the stream in front of the collective
            L.check(lib.nerf_allreduce_sum(L.ptr(buf), n, comm, L.stream_of(buf)), "nerf_allreduce_sum")
            torch.cuda.synchronize()
            assert torch.equal(buf, want)
        L.check(lib.nerf_allreduce_sum(None, 0, comm, L.stream_of(buf)), "nerf_allreduce_sum")   # empty bucket
        assert lib.nerf_allreduce_sum(None, 5, comm, L.stream_of(buf)) != 0
    finally:
        L.check(lib.nerf_comm_destroy(comm), "nerf_comm_destroy")
