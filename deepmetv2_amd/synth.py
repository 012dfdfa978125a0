"""Seeded synthetic PF-candidate events (feature layout of
/root/reference/model/data_loader.py:72-82: pX,pY,pT,eta,d0,dz,mass,puppiWeight,pdgId,charge,fromPV).

Generated on the CPU generator so the same seed gives the same events on every machine; callers move them.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch

_PDG = torch.tensor([211.0, -211.0, 130.0, 22.0, 11.0, -11.0, 13.0, -13.0, 1.0, 2.0])
_PDG_P = torch.tensor([0.275, 0.275, 0.12, 0.25, 0.005, 0.005, 0.005, 0.005, 0.03, 0.03])
_MASS = torch.tensor([0.1396, 0.1396, 0.0, 0.0, 0.000511, 0.000511, 0.1057, 0.1057, 0.0, 0.0])
_CHG = torch.tensor([1.0, -1.0, 0.0, 0.0, -1.0, 1.0, -1.0, 1.0, 0.0, 0.0])


def make_events(sizes: Sequence[int], seed: int = 1234, device: Optional[torch.device] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns x[N,11] f32, y[B,11] f32, batch[N] i64, ptr[B+1] i64 for events with the given node counts."""
    g = torch.Generator().manual_seed(int(seed))
    sizes = [int(s) for s in sizes]
    B, N = len(sizes), int(sum(sizes))
    pt = (-2.0 * torch.log1p(-torch.rand(N, generator=g))).clamp(0.01, 500.0)
    phi = (torch.rand(N, generator=g) * 2.0 - 1.0) * math.pi
    eta = (torch.rand(N, generator=g) * 2.0 - 1.0) * 5.0
    d0 = (torch.randn(N, generator=g) * 0.1).clamp(-5000.0, 5000.0)
    dz = (torch.randn(N, generator=g) * 0.1).clamp(-5000.0, 5000.0)
    kind = torch.multinomial(_PDG_P, N, replacement=True, generator=g)
    u = torch.rand(N, generator=g)
    puppi = torch.rand(N, generator=g)
    puppi = torch.where(u < 0.3, torch.zeros_like(puppi), torch.where(u > 0.7, torch.ones_like(puppi), puppi))
    from_pv = torch.randint(0, 4, (N,), generator=g).float()
    x = torch.stack([pt * torch.cos(phi), pt * torch.sin(phi), pt, eta, d0, dz, _MASS[kind], puppi, _PDG[kind],
                     _CHG[kind], from_pv], dim=1).contiguous()
    y = torch.randn(B, 11, generator=g) * 30.0
    counts = torch.tensor(sizes, dtype=torch.int64)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    batch = torch.repeat_interleave(torch.arange(B, dtype=torch.int64), counts)
    if device is not None:
        x, y, batch, ptr = x.to(device), y.to(device), batch.to(device), ptr.to(device)
    return x, y, batch, ptr


def ragged_sizes(num_events: int, lo: int, hi: int, seed: int = 1234) -> list:
    g = torch.Generator().manual_seed(int(seed) + 7919)
    return torch.randint(lo, hi + 1, (num_events,), generator=g).tolist()
