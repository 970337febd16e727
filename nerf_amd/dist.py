"""Ray-sharded data parallelism: one process per GPU, ``torch.distributed`` with
the ``nccl`` backend (RCCL over xGMI on ROCm); ``gloo`` works for CPU rehearsal.

The reference is single-process.  Rays are independent, so the only exchange of
a training step is the parameter-gradient average: every rank renders its own
ray batch with replicated parameters, the gradients are summed across ranks and
every rank applies the same optimiser update.  Inference needs only a final
gather of the rendered rows.

Gradient layout: the parameters are grouped into buckets (one per model: 2.4 MB
for a vanilla MLP; hash tables 41.7 MB and the small matrices of the hash-grid
model) and each bucket's gradients live in ONE flat buffer, ``p.grad`` being
views into it.  A step is then: one memset per bucket, backward passes that
accumulate in place (any number of ray chunks), and one all-reduce per bucket
on the buffer itself -- no copy-in, no divide pass, no copy-out.  The all-reduce
of a bucket is issued from a gradient hook the moment the last chunk's backward
has delivered that bucket, so with a coarse and a fine model the fine model's
exchange runs under the coarse model's backward.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of n items for this rank (sizes differ by at most 1)."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class GradBuckets:
    """Flat gradient storage for groups of parameters, with the cross-rank sum.

    ``buckets``: a list of parameter lists (or one flat list = one bucket).  Works without
    ``torch.distributed`` too (world size 1: the flat storage and single memset remain).

        gb = GradBuckets([list(fine.parameters()), list(coarse.parameters())])
        gb.zero_grad()                # instead of optimizer.zero_grad()
        ... backward passes of all but the last ray chunk ...
        gb.arm()                      # the next backward completes the gradients
        last_loss.backward()          # hooks launch each bucket's all-reduce as it completes
        gb.finish()                   # wait; p.grad now holds the sum over ranks
    Scale the loss by ``1 / world`` (``grad_scale``) to get the average.
    """

    def __init__(self, buckets: Sequence, group=None, overlap: bool = True, always_reduce: bool = False):
        """``overlap=False`` issues every bucket's all-reduce from ``finish()`` (the calling thread) instead of
        from the gradient hooks: no overlap with backward, the plainest use of the process group.
        ``always_reduce=True`` issues the all-reduces in a one-rank group too (a sum over one rank): the whole
        exchange path, collectives included, can then be exercised on a single GPU."""
        self.overlap = overlap
        self.always_reduce = always_reduce
        if buckets and isinstance(buckets[0], torch.nn.Parameter):
            buckets = [list(buckets)]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets: List[List[torch.nn.Parameter]] = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        self.flats: List[torch.Tensor] = []
        self.views: List[List[torch.Tensor]] = []
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            # every view starts on a 256-byte boundary: the optimiser and clipping kernels read 16 bytes per lane
            starts, off = [], 0
            for p in b:
                starts.append(off)
                off += (p.numel() + 63) // 64 * 64
            flat = torch.zeros(off, device=b[0].device, dtype=torch.float32)
            views = []
            for p, o in zip(b, starts):
                views.append(flat[o:o + p.numel()].view_as(p))
                self._hooks.append(p.register_post_accumulate_grad_hook(
                    lambda _p, bi=bi, pi=len(views) - 1: self._delivered(bi, pi)))
            self.flats.append(flat)
            self.views.append(views)
        self._armed = False
        self._got: List[set] = [set() for _ in self.buckets]   # parameters of the bucket whose gradient is complete
        self._pending = [0] * len(self.buckets)                # gradient deliveries announced by fused ops (expect) and not yet made (done)
        self._work: List[Optional[object]] = [None] * len(self.buckets)
        self._sent = [False] * len(self.buckets)
        # parameter storage -> (bucket, index in bucket, view): the gradient sink of the fused ops (below)
        self._slot = {p.data_ptr(): (bi, pi, v) for bi, (b, views) in enumerate(zip(self.buckets, self.views))
                      for pi, (p, v) in enumerate(zip(b, views)) if p.numel()}

    # ---- gradient sink of the fused MLP ops (nerf_amd.ops.grad_sink): their backward kernels accumulate straight
    # into the views instead of handing autograd a gradient to add.
    def targets(self, params) -> Optional[List[torch.Tensor]]:
        """The views to accumulate the gradients of ``params`` into, or None unless every one of them lives here
        with its view installed as ``p.grad`` (``zero_grad()`` does that and zeroes them)."""
        out = []
        for p in params:
            slot = self._slot.get(p.data_ptr()) if p.numel() else None
            g = p.grad
            if slot is None or g is None or g.data_ptr() != slot[2].data_ptr() or g.shape != p.shape:
                return None
            out.append(slot[2])
        return out

    def _slots_of(self, params):
        return [self._slot[p.data_ptr()] for p in params if p.numel() and p.data_ptr() in self._slot]

    def expect(self, params) -> None:
        """A fused op has run forward on ``params`` and will deliver their gradients through ``done`` (possibly in
        several parts): a bucket is not complete before every delivery announced for it has arrived (one model may
        serve two passes of a render)."""
        for bi, _, _ in self._slots_of(params):
            self._pending[bi] += 1

    def done(self, params) -> None:
        """The kernels that accumulate into the views of ``params`` are enqueued -- what the post-accumulate hooks
        report when autograd does the accumulation."""
        touched = set()
        for bi, pi, _ in self._slots_of(params):
            self._pending[bi] = max(0, self._pending[bi] - 1)
            touched.add(bi)
            if self._armed:
                self._got[bi].add(pi)
        if self._armed:
            for bi in touched:
                self._maybe_send(bi)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def zero_grad(self) -> None:
        """One memset per bucket; (re)installs ``p.grad`` as views of the flat buffers."""
        for flat, b, views in zip(self.flats, self.buckets, self.views):
            flat.zero_()
            for p, v in zip(b, views):
                if p.grad is not v:
                    p.grad = v
        self._armed = False
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._sent = [False] * len(self.buckets)

    def arm(self) -> None:
        """Announce that the next backward pass completes the gradients of this step."""
        self._armed = True
        # empty parameters have nothing to wait for
        self._got = [{pi for pi, p in enumerate(b) if p.numel() == 0} for b in self.buckets]

    def _send(self, bi: int) -> None:
        self._sent[bi] = True
        if self.world > 1 or (self.always_reduce and dist.is_available() and dist.is_initialized()):
            flat = self.flats[bi]
            if flat.is_cuda and dist.get_backend(self.group) == "gloo":
                # rehearsal of the multi-GPU path with ranks sharing a GPU: gloo's staging copy of a CUDA tensor
                # was seen to wait forever when kernels are still pending on the tensor's stream
                torch.cuda.current_stream(flat.device).synchronize()
            self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _delivered(self, bi: int, pi: int) -> None:
        if not self._armed:
            return
        self._got[bi].add(pi)
        self._maybe_send(bi)

    def _maybe_send(self, bi: int) -> None:
        if (self._armed and self.overlap and not self._sent[bi] and self._pending[bi] == 0
                and len(self._got[bi]) == len(self.buckets[bi])):
            self._send(bi)

    def finish(self) -> None:
        """Send what no hook has sent (parameters the last backward did not reach), wait for all."""
        for bi in range(len(self.buckets)):
            if not self._sent[bi]:
                self._send(bi)
        for w in self._work:
            if w is not None:
                w.wait()
        self._armed = False

    def parameters(self) -> List[torch.nn.Parameter]:
        return [p for b in self.buckets for p in b]

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def accumulate_step(n_items: int, chunk: int, loss_of_chunk: Callable[[int, int], torch.Tensor],
                    buckets: Optional[GradBuckets] = None, grad_scale: Optional[float] = None) -> torch.Tensor:
    """Forward + backward over ``n_items`` rays in chunks of at most ``chunk``, accumulating the parameter
    gradients, so that activation memory is bounded by one chunk -- the counterpart of the reference's chunk
    loop (src/train_nerf_llff.py:50-56) and of its halve-on-OOM retry (src/train_nerf.py:34-44), moved from
    the model call to the whole step.

    ``loss_of_chunk(lo, hi)`` returns a loss that is a MEAN over rays lo..hi; it is weighted by
    (hi - lo) / n_items, so the accumulated gradient is that of the mean over all rays.  With ``buckets`` the
    gradients land in its flat buffers and the cross-rank sum starts under the last chunk's backward;
    ``grad_scale`` defaults to 1 / world size.  Returns the detached total loss (a 0-d tensor, no host sync).
    """
    import contextlib
    if grad_scale is None:
        grad_scale = buckets.grad_scale if buckets is not None else 1.0
    sink = contextlib.nullcontext()
    if buckets is not None:
        buckets.zero_grad()
        if any(f.is_cuda for f in buckets.flats):
            # the fused ops' backward kernels accumulate straight into the flat buffers (no per-parameter
            # zero-fill and add by autograd)
            from . import ops
            sink = ops.grad_sink(buckets)
    total = None
    spans = [(lo, min(lo + chunk, n_items)) for lo in range(0, n_items, chunk)]
    with sink:
        for k, (lo, hi) in enumerate(spans):
            loss = loss_of_chunk(lo, hi)
            if hi - lo != n_items:
                loss = loss * ((hi - lo) / n_items)
            if buckets is not None and k == len(spans) - 1:
                buckets.arm()
            # the root gradient is a constant kept per device: left to autograd it is a ones_like() fill per backward
            (loss * grad_scale if grad_scale != 1.0 else loss).backward(gradient=_one(loss))
            total = loss.detach() if total is None else total + loss.detach()
    if buckets is not None:
        buckets.finish()
    return total


_ONES = {}


def _one(like: torch.Tensor) -> torch.Tensor:
    if like.dim() != 0:
        return torch.ones_like(like)
    key = (like.device, like.dtype)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((), device=like.device, dtype=like.dtype)
    return t


class GradAverager:
    """Averages ``p.grad`` of the given parameters over all ranks with a single flat all-reduce, for
    gradients that already exist as ordinary tensors (``GradBuckets`` avoids the copies).  Parameters
    whose grad is None on this rank contribute zeros."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def average(self) -> None:
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
        have = [(p.grad, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        if have:
            torch._foreach_copy_([v for _, v in have], [g for g, _ in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def gather_rows(local: torch.Tensor, sizes: List[int], group=None) -> torch.Tensor:
    """Concatenate per-rank row blocks [n_r, ...] (n_r = sizes[r]) on every rank:
    the image gather that closes a sharded inference render."""
    world = dist.get_world_size(group)
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
