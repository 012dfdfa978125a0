"""The optimizer of the reference's training loop (/root/reference/train.py:75 `optim.AdamW(model.parameters(), lr=...)`,
stepped at train.py:52) for the flat parameter tensor of `parallel.FlatModule`: one HIP launch per step
(`dmet_adamw_f32`) where `torch.optim.AdamW(fused=True)` takes two (its multi-tensor kernel, 15 us for 6 641
parameters, and the step-counter increment).  Same update rule, same hyper-parameters and defaults, same state names
(`step`, `exp_avg`, `exp_avg_sq`, plus `bias_pow` = beta^step as running products in double); the step state lives on
the device, so the step replays inside a hipGraph.
"""
from __future__ import annotations

from typing import Iterable

import torch

from . import _lib, _native


class FlatAdamW(torch.optim.Optimizer):
    """AdamW (decoupled weight decay, no amsgrad, no maximize) over contiguous fp32 GPU parameters."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid AdamW hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                        and g.dtype == torch.float32):
                    raise TypeError("FlatAdamW: parameters and gradients must be contiguous float32 GPU tensors")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["bias_pow"] = torch.ones(2, dtype=torch.float64, device=p.device)   # beta1^step, beta2^step
                with _native._on(p.device):
                    _lib.check(L.dmet_adamw_f32(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                                                st["bias_pow"].data_ptr(), p.numel(),
                                                float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                float(group["weight_decay"]), _native._stream(p.device)),
                               "dmet_adamw_f32")
        return loss
