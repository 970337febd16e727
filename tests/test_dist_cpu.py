"""Ray-sharded data parallelism helpers, rehearsed with two gloo ranks on the CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_amd.dist import GradAverager, GradBuckets, accumulate_step, broadcast_parameters, gather_rows, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        broadcast_parameters(model, src=0)
        params = list(model.parameters())
        # every rank owns a slice of a global batch of "rays"
        g = torch.Generator().manual_seed(5)
        x_all, y_all = torch.randn(20, 7, generator=g), torch.randn(20, 3, generator=g)
        b, e = shard_range(20, rank, world)
        loss = torch.nn.functional.mse_loss(model(x_all[b:e]), y_all[b:e], reduction="sum") / 20
        loss.backward()
        params[-1].grad = None if rank == 1 else params[-1].grad   # a rank without a gradient contributes zeros
        expect_last = None
        avg = GradAverager(params)
        avg.average()
        # reference: the full-batch gradient on one process (sum over ranks, the averager divides by world)
        ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        ref.load_state_dict(model.state_dict())
        (torch.nn.functional.mse_loss(ref(x_all), y_all, reduction="sum") / 20).backward()
        ok = True
        for i, (p, q) in enumerate(zip(params[:-1], list(ref.parameters())[:-1])):
            ok &= torch.allclose(p.grad * world, q.grad, atol=1e-6)
        ok &= params[-1].grad is not None
        rows = torch.full((e - b, 3), float(rank))
        img = gather_rows(rows, [shard_range(20, r, world)[1] - shard_range(20, r, world)[0] for r in range(world)])
        ok &= img.shape == (20, 3) and float(img[:b].sum()) == 0.0 * b and float(img[-1, 0]) == world - 1
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _chunked_worker(rank, world, port, out):
    """The chunked training step (ray chunks, gradients accumulated in flat bucket buffers, one all-reduce per
    bucket issued from the gradient hooks of the last chunk) against a single-process full-batch gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7)
        coarse = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 3))
        fine = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
        unused = torch.nn.Parameter(torch.ones(5))          # never reached by backward: finish() must still send it
        buckets = GradBuckets([list(fine.parameters()), list(coarse.parameters()) + [unused]])
        g = torch.Generator().manual_seed(5)
        n_all = 46                                          # rays of the global batch; 23 per rank, chunks of 8 (ragged tail)
        x_all, y_all = torch.randn(n_all, 6, generator=g), torch.randn(n_all, 3, generator=g)
        b, e = shard_range(n_all, rank, world)
        x, y = x_all[b:e], y_all[b:e]

        def loss_of_chunk(lo, hi):                          # a mean over the chunk's rays, like nerf_loss
            return torch.nn.functional.mse_loss(coarse(x[lo:hi]), y[lo:hi]) + torch.nn.functional.mse_loss(fine(x[lo:hi]), y[lo:hi])

        ok = True
        for step in range(2):                               # twice: the buffers and hooks are reused
            loss = accumulate_step(e - b, 8, loss_of_chunk, buckets)
            ptrs = [p.grad.data_ptr() for p in buckets.parameters()]
            ok &= ptrs == [v.data_ptr() for views in buckets.views for v in views]   # accumulated in place
            ref_c = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 3))
            ref_f = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
            ref_c.load_state_dict(coarse.state_dict())
            ref_f.load_state_dict(fine.state_dict())
            full = torch.nn.functional.mse_loss(ref_c(x_all), y_all) + torch.nn.functional.mse_loss(ref_f(x_all), y_all)
            full.backward()
            for p, q in zip(list(fine.parameters()) + list(coarse.parameters()),
                            list(ref_f.parameters()) + list(ref_c.parameters())):
                ok &= torch.allclose(p.grad, q.grad, atol=1e-6)
            ok &= float(unused.grad.abs().sum()) == 0.0
            # the per-rank losses average to the full-batch loss (equal shard sizes)
            t = loss.clone()
            dist.all_reduce(t)
            ok &= abs(float(t) / world - float(full)) < 1e-5
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_chunked_step_with_bucketed_allreduce_two_ranks():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_chunked_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_grad_buckets_single_process():
    """World size 1 (no process group): flat storage, in-place accumulation over chunks, no exchange."""
    torch.manual_seed(1)
    m = torch.nn.Linear(4, 2)
    gb = GradBuckets(list(m.parameters()))
    x, y = torch.randn(10, 4), torch.randn(10, 2)
    loss = accumulate_step(10, 3, lambda lo, hi: torch.nn.functional.mse_loss(m(x[lo:hi]), y[lo:hi]), gb)
    ref = torch.nn.Linear(4, 2)
    ref.load_state_dict(m.state_dict())
    full = torch.nn.functional.mse_loss(ref(x), y)
    full.backward()
    assert abs(float(loss) - float(full)) < 1e-6
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6)
    assert m.weight.grad.data_ptr() == gb.flats[0].data_ptr()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 4096, 640000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_grad_average_and_gather_two_ranks():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_gradient_sink_waits_for_every_op_call_of_a_bucket():
    """The fused ops deliver gradients straight into the bucket views and report through expect()/done().  A model
    that serves two passes of a render reports twice: its bucket is complete after the second report, not the
    first; a bucket none of whose parameters was touched is left to finish()."""
    from nerf_amd.dist import GradBuckets
    a = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(0, 4))]
    b = [torch.nn.Parameter(torch.zeros(2, 2))]
    gb = GradBuckets([a, b])
    gb.zero_grad()
    assert [t.data_ptr() for t in gb.targets(a[:2])] == [v.data_ptr() for v in gb.views[0][:2]]
    assert gb.targets([torch.nn.Parameter(torch.zeros(3))]) is None          # a stranger: the sink declines
    # an earlier chunk: forward, backward, nothing armed
    gb.expect(a)
    gb.done(a)
    assert gb._sent == [False, False] and gb._pending == [0, 0]
    # the last chunk: two op calls on the same parameters, the second delivering in two parts
    gb.expect(a)
    gb.expect(a)
    gb.arm()
    gb.done(a)
    assert gb._sent == [False, False]
    gb.done(a[:1])
    assert gb._sent == [False, False]
    gb.done(a[1:])
    assert gb._sent == [True, False]
    gb.finish()
    assert gb._sent == [True, True]
    # the next step starts clean, and a view that is no longer installed as p.grad makes the sink decline
    gb.zero_grad()
    assert gb._pending == [0, 0] and gb._sent == [False, False]
    a[0].grad = torch.zeros(5, 3)
    assert gb.targets(a[:2]) is None
