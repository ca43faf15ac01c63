"""Optimizers of the reference's search space (utils/training_models_multimodal.py:318-325; re-exported by
BIOINF_tesi/models/utils/optim/__init__.py: Adam, RMSprop from torch, Nadam from timm) as fused HIP kernels.

They are ``torch.optim.Optimizer`` subclasses, so the harness' ``optimizer.zero_grad()/step()`` calls work
unchanged.  One launch per parameter tensor updates parameter + moments (+ the bf16 shadow the EMB_BF16
kernels read); the step count lives on the device so a captured step replays correctly.
Semantics: coupled L2 weight decay as in torch.optim; Adam/RMSprop pinned by fixture G7; Nadam follows
timm's published algorithm -- timm is not installed here, so Nadam is PARITY UNPINNED.
"""
import torch

from . import _lib
from ._lib import DTYPE_CODE, check, ptr, stream


class _Fused(torch.optim.Optimizer):
    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._step_dev = {}

    def _counter(self, device):
        if device not in self._step_dev:
            self._step_dev[device] = torch.zeros(1, dtype=torch.int64, device=device)
        return self._step_dev[device]

    def _tick(self, devices):
        for d in devices:
            check(_lib.lib().emb_counter_add(ptr(self._counter(d)), 1, stream()), "emb_counter_add")

    def _params(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype not in (torch.float32, torch.float64):
                    raise TypeError("fused optimizers update fp32 / fp64 master parameters")
                _lib.require_cuda(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                yield group, p, g

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items = list(self._params())
        self._tick({p.device for _, p, _ in items})
        for group, p, g in items:
            self._update(group, p, g, self.state[p])
        return loss

    def shadow_of(self, p):
        """bf16 copy of `p` maintained by the update kernel (created on first use)."""
        st = self.state[p]
        if "shadow" not in st:
            st["shadow"] = p.detach().to(torch.bfloat16)
        return st["shadow"]


class Adam(_Fused):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _update(self, group, p, g, st):
        if "exp_avg" not in st:
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
        b1, b2 = group["betas"]
        check(_lib.lib().emb_adam_step(ptr(p), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(st.get("shadow")),
                                       p.numel(), group["lr"], b1, b2, group["eps"], group["weight_decay"], 0,
                                       ptr(self._counter(p.device)), DTYPE_CODE[p.dtype], stream()), "emb_adam_step")


class RMSprop(_Fused):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))

    def _update(self, group, p, g, st):
        if "square_avg" not in st:
            st["square_avg"] = torch.zeros_like(p)
        check(_lib.lib().emb_rmsprop_step(ptr(p), ptr(g), ptr(st["square_avg"]), ptr(st.get("shadow")), p.numel(),
                                          group["lr"], group["alpha"], group["eps"], group["weight_decay"],
                                          DTYPE_CODE[p.dtype], stream()), "emb_rmsprop_step")


class Nadam(_Fused):
    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, schedule_decay=4e-3):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      schedule_decay=schedule_decay))

    def _update(self, group, p, g, st):
        if "exp_avg" not in st:
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
            st["m_schedule"] = torch.ones(2, dtype=torch.float64, device=p.device)
        b1, b2 = group["betas"]
        check(_lib.lib().emb_nadam_step(ptr(p), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(st["m_schedule"]),
                                        ptr(st.get("shadow")), p.numel(), group["lr"], b1, b2, group["eps"],
                                        group["weight_decay"], group["schedule_decay"], 0,
                                        ptr(self._counter(p.device)), DTYPE_CODE[p.dtype], stream()), "emb_nadam_step")


__all__ = ["Adam", "RMSprop", "Nadam"]
