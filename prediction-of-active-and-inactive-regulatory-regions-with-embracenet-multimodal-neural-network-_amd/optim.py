"""Optimizers of the reference's search space (utils/training_models_multimodal.py:318-325; re-exported by
BIOINF_tesi/models/utils/optim/__init__.py: Adam, RMSprop from torch, Nadam from timm) as fused HIP kernels.

They are ``torch.optim.Optimizer`` subclasses, so the harness' ``optimizer.zero_grad()/step()`` calls work
unchanged.  ONE launch per step updates every parameter tensor of a dtype (multi-tensor kernel: parameter +
moments + the bf16 shadow the EMB_BF16 kernels read); the step count lives on the device so a captured step
replays correctly.  Semantics: coupled L2 weight decay as in torch.optim; Adam/RMSprop pinned by fixture G7;
Nadam follows timm's published algorithm -- timm is not installed here, so Nadam is PARITY UNPINNED.
"""
import ctypes

import torch

from . import _lib
from . import functional as F_
from ._lib import DTYPE_CODE, check, ptr, stream


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class _Fused(torch.optim.Optimizer):
    state_names = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._step_dev = {}

    def _counter(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:      # "cuda" and "cuda:0" must name the same counter
            device = torch.device("cuda", torch.cuda.current_device())
        if device not in self._step_dev:
            self._step_dev[device] = torch.zeros(1, dtype=torch.int64, device=device)
        return self._step_dev[device]

    def step_counter(self, device):
        """int64 device scalar holding the number of updates done; `step` advances it before updating unless
        ``external_tick`` is set (then functional.weighted_ce_with_grad(ticks=...) does, once per loss evaluation)."""
        return self._counter(device)

    def _init_state(self, p, st):
        for name in self.state_names:
            if name not in st:
                st[name] = torch.zeros_like(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ticked = set()
        for group in self.param_groups:
            buckets = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype not in (torch.float32, torch.float64):
                    raise TypeError("fused optimizers update fp32 / fp64 master parameters")
                _lib.require_cuda(p)
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                self._init_state(p, self.state[p])
                buckets.setdefault((p.device, p.dtype), []).append(p)
            for (device, dtype), plist in buckets.items():
                if device not in ticked:        # one tick per device per step, before any update of this step
                    if not getattr(self, "external_tick", False):   # else: advanced by the loss kernel (step_counter)
                        check(_lib.lib().emb_counter_add(ptr(self._counter(device)), 1, stream()), "emb_counter_add")
                    ticked.add(device)
                self._launch(group, plist, device, dtype)
        return loss

    def _tables(self, plist):
        params = _ptr_array(plist)
        grads = _ptr_array([p.grad for p in plist])
        states = [_ptr_array([self.state[p][n] for p in plist]) for n in self.state_names]
        shadows = _ptr_array([F_.shadow_lookup(p) for p in plist])
        sizes = (ctypes.c_int64 * len(plist))(*[p.numel() for p in plist])
        return params, grads, states, shadows, sizes


class Adam(_Fused):
    state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _launch(self, group, plist, device, dtype):
        params, grads, (m, v), shadows, sizes = self._tables(plist)
        b1, b2 = group["betas"]
        check(_lib.lib().emb_adam_step_multi(params, grads, m, v, shadows, sizes, len(plist), group["lr"], b1, b2,
                                             group["eps"], group["weight_decay"], 0, ptr(self._counter(device)),
                                             DTYPE_CODE[dtype], stream()), "emb_adam_step_multi")


class RMSprop(_Fused):
    state_names = ("square_avg",)

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))

    def _launch(self, group, plist, device, dtype):
        params, grads, (sq,), shadows, sizes = self._tables(plist)
        check(_lib.lib().emb_rmsprop_step_multi(params, grads, sq, shadows, sizes, len(plist), group["lr"], group["alpha"],
                                                group["eps"], group["weight_decay"], DTYPE_CODE[dtype], stream()),
              "emb_rmsprop_step_multi")


class Nadam(_Fused):
    state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, schedule_decay=4e-3):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      schedule_decay=schedule_decay))
        self._m_schedule = {}

    def _launch(self, group, plist, device, dtype):
        key = (id(group), device, dtype)
        if key not in self._m_schedule:       # timm keeps m_schedule per parameter; all copies evolve identically
            self._m_schedule[key] = torch.ones(2, dtype=torch.float64, device=device)
        params, grads, (m, v), shadows, sizes = self._tables(plist)
        b1, b2 = group["betas"]
        check(_lib.lib().emb_nadam_step_multi(params, grads, m, v, ptr(self._m_schedule[key]), shadows, sizes, len(plist),
                                              group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                              group["schedule_decay"], 0, ptr(self._counter(device)), DTYPE_CODE[dtype],
                                              stream()), "emb_nadam_step_multi")


__all__ = ["Adam", "RMSprop", "Nadam"]
