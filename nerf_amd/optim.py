"""Adam as one HIP launch over all parameters.

Drop-in for the way the reference's training loops use ``torch.optim.Adam``
(src/train_nerf_llff.py:82 ``Adam(lr=5e-4)``; src/train_nerf_ngp.py:55
``Adam(lr=1e-2, eps=1e-15)``): same constructor arguments for the options those
loops use, same update rule, and a ``state_dict`` with torch's keys
(``step`` / ``exp_avg`` / ``exp_avg_sq``), so checkpoints move both ways.  ``weight_decay`` is torch's L2
form (src/train_nerf_llff.py:82, src/train_nerf_skow.py:47 use 1e-6); amsgrad and maximize are
not implemented and raise.  ``clip_grad_norm_`` (src/train_nerf.py:165) lives in ``nerf_amd.ops``.

``capturable=True`` (torch's name for it) keeps the step count AND the learning rate in device memory, so that a
whole training step -- this optimiser's launch included -- can be captured into a HIP graph and replayed
(``nerf_amd.graph.GraphedStep``): ``state[p]["step"]`` is then a float32 scalar on the GPU, as with
``torch.optim.Adam(capturable=True)``, shared by the parameters of a group.  ``group["lr"]`` stays the Python
float that ``torch.optim.lr_scheduler.*`` reads and writes (ExponentialLR src/train_nerf.py:87,204, StepLR
src/train_nerf_llff.py:83, ReduceLROnPlateau / CosineAnnealingLR src/train_nerf_ngp.py:57-62); ``sync_lr()`` -- called
by every eager ``step()`` and by ``GraphedStep`` in front of every replay -- sends it to the device when it changed.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, maximize: bool = False, capturable: bool = False):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdam implements Adam without amsgrad or maximize")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, capturable=bool(capturable)))
        self._dev_state = {}   # capturable groups: index -> 4 device floats, [0] = steps taken, [3] = learning rate
        self._dev_lr = {}      # capturable groups: index -> the rate the device holds

    def _capturable_state(self, gi: int, ps) -> torch.Tensor:
        """The group's step state in device memory; every parameter's ``state["step"]`` is the same 0-d view of
        its first float.  Built on the first step (and after ``load_state_dict``) from the steps the states hold."""
        entry = self._dev_state.get(gi)
        if entry is None:
            steps = {float(self.state[p]["step"]) for p in ps} or {0.0}
            if len(steps) != 1:
                raise L.NerfHipError("FusedAdam(capturable=True) needs one step count per parameter group")
            dev = torch.zeros(4, device=ps[0].device, dtype=torch.float32)
            dev[0] = steps.pop()
            entry = self._dev_state[gi] = (dev, dev[0])
        dev, view = entry
        for p in ps:
            if self.state[p]["step"] is not view:
                self.state[p]["step"] = view
        return dev

    def sync_lr(self) -> None:
        """Send every capturable group's ``lr`` to its device state if it differs from what the device holds (one
        one-thread launch per changed group on the current stream, nothing when unchanged).  Under stream capture
        nothing may be written -- the value would be replayed with the graph -- so a rate that changed since the last
        eager step raises there instead of training on silently with the old one."""
        for gi, group in enumerate(self.param_groups):
            entry = self._dev_state.get(gi)
            if entry is None or not group.get("capturable", False):
                continue
            lr = float(group["lr"])
            if self._dev_lr.get(gi) == lr:
                continue
            dev = entry[0]
            if torch.cuda.is_current_stream_capturing():
                raise L.NerfHipError("FusedAdam: the learning rate changed between the last eager step and a stream "
                                     "capture; call sync_lr() before capturing")
            L.check(L.lib().nerf_adam_set_lr(L.ptr(dev), lr, L.stream_of(dev)), "nerf_adam_set_lr")
            self._dev_lr[gi] = lr

    def state_dict(self):
        """torch's layout.  The parameters of a capturable group share one step tensor; the dict gets a copy per
        parameter, so that an optimiser which advances each parameter's step by itself (this class without
        ``capturable``, ``torch.optim.Adam``) can load it."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (v.clone() if n == "step" and torch.is_tensor(v) else v) for n, v in st.items()}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._dev_state = {}   # rebuilt from the loaded step counts on the next step
        self._dev_lr = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.NerfHipError("FusedAdam needs contiguous float32 parameters")
                if p.grad.is_sparse:
                    raise L.NerfHipError("FusedAdam does not take sparse gradients")
                L.need_gpu(p)
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if group.get("capturable", False):
                # one launch pair for the group, nothing read back, no host-side step count
                dev = self._capturable_state(gi, ps)
                self.sync_lr()
                n = len(ps)
                numel = (C.c_int64 * n)(*[p.numel() for p in ps])
                L.check(L.lib().nerf_adam_step_capturable(
                    n, L.ptr_array(ps), L.ptr_array([L.f32c(p.grad) for p in ps]),
                    L.ptr_array([self.state[p]["exp_avg"] for p in ps]),
                    L.ptr_array([self.state[p]["exp_avg_sq"] for p in ps]), numel, L.ptr(dev),
                    group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                    L.stream_of(ps[0])), "nerf_adam_step_capturable")
                continue
            # parameters of one group share the step count unless states were loaded piecemeal
            by_step = {}
            for p in ps:
                by_step.setdefault(int(self.state[p]["step"].item()), []).append(p)
            for step0, plist in by_step.items():
                grads = [L.f32c(p.grad) for p in plist]
                n = len(plist)
                numel = (C.c_int64 * n)(*[p.numel() for p in plist])
                L.check(L.lib().nerf_adam_step(
                    n, L.ptr_array(plist), L.ptr_array(grads), L.ptr_array([self.state[p]["exp_avg"] for p in plist]),
                    L.ptr_array([self.state[p]["exp_avg_sq"] for p in plist]), numel, step0 + 1, group["lr"],
                    group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                    L.stream_of(plist[0])), "nerf_adam_step")
                for p in plist:
                    self.state[p]["step"] += 1
        return loss
