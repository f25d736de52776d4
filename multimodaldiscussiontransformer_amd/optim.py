"""Optimiser + LR schedule of the reference launch (FairSeq `adam` + `polynomial_decay`,
mDT/experiments/hateful_discussions/run_train.sh:38-40), fused: one HIP kernel per parameter
tensor reads the fp32 gradient arena and updates moments, fp32 master weights and the
working-precision parameter in a single pass."""
from __future__ import annotations

import torch

from ._lib import check, dt, lib, ptr, stream


class FusedAdam:
    def __init__(self, params, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, multi_tensor=True):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.multi_tensor = multi_tensor          # False: one launch per tensor (tests compare the two)
        self.state = {}
        for p in self.params:
            st = dict(m=torch.zeros(p.shape, dtype=torch.float32, device=p.device),
                      v=torch.zeros(p.shape, dtype=torch.float32, device=p.device))
            if p.dtype != torch.float32:
                st["master"] = p.detach().float().clone()
            self.state[id(p)] = st

    def _tables(self):
        """Device tables for the one-launch update, built once: parameters of one dtype whose gradients live in the
        fp32 ``main_grad`` arena (stable addresses).  Others fall back to one launch per tensor."""
        if not self.multi_tensor:
            return {}
        if getattr(self, "_tab", None) is not None:
            return self._tab
        import numpy as np
        groups = {}
        for p in self.params:
            g = getattr(p, "main_grad", None)
            if g is None or not g.is_contiguous() or not p.data.is_contiguous():
                continue
            groups.setdefault(p.dtype, []).append(p)
        tab = {}
        for dtype, ps in groups.items():
            rec = np.zeros((len(ps), 6), dtype=np.int64)
            first = np.zeros(len(ps) + 1, dtype=np.int64)
            for i, p in enumerate(ps):
                st = self.state[id(p)]
                rec[i] = (p.data.data_ptr(), st["master"].data_ptr() if "master" in st else 0, p.main_grad.data_ptr(),
                          st["m"].data_ptr(), st["v"].data_ptr(), p.numel())
                first[i + 1] = first[i] + (p.numel() + 4095) // 4096
            dev = ps[0].device
            tab[dtype] = (torch.from_numpy(rec).to(dev), torch.from_numpy(first).to(dev), len(ps), int(first[-1]),
                          {id(p) for p in ps}, ps, [p.main_grad.data_ptr() for p in ps])
        self._tab = tab
        return tab

    def step(self, lr=None, grad_scale: torch.Tensor = None):
        """Gradients are read from ``p.main_grad`` (fp32 arena) or ``p.grad``; ``grad_scale`` is an optional
        fp32 device scalar multiplied into every gradient."""
        self.step_count += 1
        lr = self.lr if lr is None else lr
        done = set()
        # main_grad views move when the DDP arena is re-laid-out after the first step: rebuild the tables then
        if getattr(self, "_tab", None) is not None and any(
                [p.main_grad.data_ptr() for p in ps] != gptrs for (_, _, _, _, _, ps, gptrs) in self._tab.values()):
            self._tab = None
        for dtype, (rec, first, n, total, ids, _, _) in self._tables().items():
            check(lib.mdt_adam_step_multi(stream(), 0 if dtype == torch.float32 else 1, n, ptr(rec), ptr(first), total,
                                          float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                          float(self.weight_decay), self.step_count, ptr(grad_scale)), "mdt_adam_step_multi")
            done |= ids
        self._step_rest(lr, grad_scale, done)
        from . import fp8
        if fp8.ACTIVE is not None:
            fp8.ACTIVE.optimizer_stepped()      # the cached 8-bit weight copies are stale now
        from . import engine
        engine.weights_changed()                # ... and the transposed copies the input gradients read

    def _step_rest(self, lr, grad_scale, done):
        for p in self.params:
            if id(p) in done:
                continue
            g = getattr(p, "main_grad", None)
            if g is None:
                g = p.grad
                if g is None:
                    continue
                if g.dtype != torch.float32:
                    g = g.float()
            st = self.state[id(p)]
            check(lib.mdt_adam_step(stream(), dt(p), p.numel(), ptr(p.data), ptr(st.get("master")), ptr(g.contiguous()),
                                    ptr(st["m"]), ptr(st["v"]), float(lr), float(self.betas[0]), float(self.betas[1]),
                                    float(self.eps), float(self.weight_decay), self.step_count, ptr(grad_scale)),
                  "mdt_adam_step")


class PolynomialDecayLR:
    """FairSeq ``polynomial_decay``: linear warm-up to ``lr`` over ``warmup_updates`` then
    (lr - end_lr) * (1 - progress)^power + end_lr until ``total_num_update``."""

    def __init__(self, lr=3e-5, end_lr=3e-7, warmup_updates=3246, total_num_update=10820, power=1.0):
        self.lr, self.end_lr, self.warmup, self.total, self.power = lr, end_lr, warmup_updates, total_num_update, power

    def for_update(self, k: int) -> float:
        """Learning rate update ``k`` (1-based) RUNS with under FairSeq's trainer.  FairSeq calls
        ``lr_scheduler.step_update(num_updates)`` after an update, so update k uses the rate of ``num_updates = k - 1``;
        before the first update the scheduler's constructor has set ``warmup_factor * lr`` with warmup_factor =
        1 / warmup_updates (fairseq/optim/lr_scheduler/polynomial_decay_schedule.py: __init__ and step_update)."""
        done = k - 1
        if done <= 0:
            return self.lr / self.warmup if self.warmup > 0 else self.lr
        return self(done)

    def __call__(self, num_updates: int) -> float:
        if self.warmup > 0 and num_updates <= self.warmup:
            return self.lr * num_updates / float(self.warmup)
        if num_updates >= self.total:
            return self.end_lr
        pct = 1 - (num_updates - self.warmup) / float(self.total - self.warmup)
        return (self.lr - self.end_lr) * pct ** self.power + self.end_lr
