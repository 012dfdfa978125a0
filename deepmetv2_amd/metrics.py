"""Evaluation-side metrics of the reference: /root/reference/model/net.py:64-161.

`u_perp_par_loss`, `resolution` and the `metrics` registry with the reference's names and argument order.  The per-event
MET sums are the HIP reduction (`met_reduce`); what follows is O(B) arithmetic on [B,2] vectors, kept in torch.

Recoil decomposition of a transverse vector v against the boson q_T (net.py:64-69,139-145):
    response = (v . q) / (q . q),   v_par = response * q,   u_par = |v_par| - |q|,   u_perp = |v - v_par|.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .scatter import met_reduce


def recoil_components(vec: torch.Tensor, v_qt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(u_perp, u_par, response) of the [B,2] vectors `vec` against `v_qt`."""
    qq = (v_qt * v_qt).sum(1)
    response = (vec * v_qt).sum(1) / qq
    v_par = response.unsqueeze(1) * v_qt
    u_par = v_par.norm(dim=1) - qq.sqrt()
    u_perp = (vec - v_par).norm(dim=1)
    return u_perp, u_par, response


def _met(weights: torch.Tensor, prediction: torch.Tensor, batch: torch.Tensor, ptr: Optional[torch.Tensor],
         num_events: int) -> torch.Tensor:
    return met_reduce(weights, prediction, batch, ptr=ptr, num_events=num_events)


def u_perp_par_loss(weights: torch.Tensor, prediction: torch.Tensor, truth: torch.Tensor, batch: torch.Tensor,
                    ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """net.py:70-90: 0.5 * mean(u_par^2 + u_perp^2) of the predicted MET = -(sum w px, sum w py).
    The reference builds q_T from truth column 0 for BOTH components (net.py:71-72); kept as is."""
    v_qt = torch.stack((truth[:, 0], truth[:, 0]), dim=1)
    vec = -_met(weights, prediction, batch, ptr, truth.shape[0])
    u_perp, u_par, _ = recoil_components(vec, v_qt)
    return 0.5 * (u_par ** 2 + u_perp ** 2).mean()


def resolution(weights: torch.Tensor, prediction: torch.Tensor, truth: torch.Tensor, batch: torch.Tensor,
               ptr: Optional[torch.Tensor] = None) -> Tuple[Dict[str, List[np.ndarray]], np.ndarray]:
    """net.py:92-157: per-event [u_perp, u_par, response] (numpy) of the predicted MET and of the reference MET
    flavours stored in `truth` (columns 0-1 q_T, 2-3 PF MET, 4-5 PUPPI MET, and, when present, 6-7 / 8-9 the DeepMET
    response / resolution tunes), plus |q_T| per event."""
    v_qt = truth[:, 0:2]

    def compute(vec: torch.Tensor) -> List[np.ndarray]:
        return [t.detach().cpu().numpy() for t in recoil_components(vec, v_qt)]

    out = {
        "MET": compute(-_met(weights, prediction, batch, ptr, truth.shape[0])),
        "pfMET": compute(truth[:, 2:4]),
        "puppiMET": compute(truth[:, 4:6]),
    }
    if truth.shape[1] > 6:
        out["deepMETResponse"] = compute(truth[:, 6:8])
        out["deepMETResolution"] = compute(truth[:, 8:10])
    return out, v_qt.norm(dim=1).detach().cpu().numpy()


# net.py:159-161: "maintain all metrics required in this dictionary"
metrics = {"resolution": resolution}
