"""The optimizer of the reference's training loop (/root/reference/train.py:75 `optim.AdamW(model.parameters(), lr=...)`,
stepped at train.py:52) for the flat parameter tensor of `parallel.FlatModule`: one HIP launch per step
(`dmet_adamw_f32`) where `torch.optim.AdamW(fused=True)` takes two (its multi-tensor kernel, 15 us for 6 641
parameters, and the step-counter increment).  Same update rule, same hyper-parameters and defaults, same state names
(`step`, `exp_avg`, `exp_avg_sq`, plus `bias_pow` = beta^step as running products in double and `lr_dev`, the learning
rate as a device double); the step state lives on the device, so the step replays inside a hipGraph -- including after a
scheduler (train.py:76 ReduceLROnPlateau) changed `param_groups[i]["lr"]`: `sync_hyper()` pushes the new value into
`lr_dev` (`step()` and `parallel.GraphedTrainStep` call it; one tiny fill, only when the value changed).  A state_dict
written by `torch.optim.AdamW` (`step`, `exp_avg`, `exp_avg_sq` only) loads: the missing pieces are rebuilt from `step`.
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

    def _state_of(self, p, group):
        """The device-side state of one parameter, created on first use or completed after load_state_dict."""
        st = self.state[p]
        b1, b2 = group["betas"]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not torch.is_tensor(st["step"]) or st["step"].device != p.device or st["step"].dtype != torch.float32:
            # torch.optim.AdamW keeps `step` as a CPU tensor (or a number) unless capturable=True
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32).to(p.device).reshape(())
        # (load_state_dict casts floating state tensors to the parameter's dtype: a float32 bias_pow / lr_dev is rebuilt)
        if "bias_pow" not in st or st["bias_pow"].device != p.device or st["bias_pow"].dtype != torch.float64:
            # beta^step from the step count (no host sync): a state written by torch.optim.AdamW does not carry it
            beta = torch.tensor([b1, b2], dtype=torch.float64, device=p.device)
            st["bias_pow"] = beta ** st["step"].to(torch.float64)
        if "lr_dev" not in st or st["lr_dev"].device != p.device or st["lr_dev"].dtype != torch.float64:
            st["lr_dev"] = torch.full((1,), float(group["lr"]), dtype=torch.float64, device=p.device)
            st["lr_host"] = float(group["lr"])
        return st

    @torch.no_grad()
    def sync_hyper(self) -> None:
        """Push a changed learning rate to the device scalar the kernel reads.  Call it (outside any stream capture)
        before replaying a captured step; `step()` calls it itself."""
        for group in self.param_groups:
            lr = float(group["lr"])
            for p in group["params"]:
                st = self.state.get(p)
                if st and "lr_dev" in st and st.get("lr_host") != lr:
                    st["lr_dev"].fill_(lr)
                    st["lr_host"] = lr

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
                st = self._state_of(p, group)
                if st["lr_host"] != float(group["lr"]) and not torch.cuda.is_current_stream_capturing():
                    st["lr_dev"].fill_(float(group["lr"]))
                    st["lr_host"] = float(group["lr"])
                with _native._on(p.device):
                    _lib.check(L.dmet_adamw_lr_f32(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                   st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                                                   st["bias_pow"].data_ptr(), p.numel(), st["lr_dev"].data_ptr(),
                                                   float(b1), float(b2), float(group["eps"]),
                                                   float(group["weight_decay"]), _native._stream(p.device)),
                               "dmet_adamw_lr_f32")
        return loss
