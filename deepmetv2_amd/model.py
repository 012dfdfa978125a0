"""The graph-MET model wired onto the HIP operators (rows H1/H2 of SURVEY.md section 8a).

Own counterpart of /root/reference/model/graph_met_network.py:11-69 (`GraphMETNetwork`) and
/root/reference/model/net.py:38-62 (`Net`, `loss_fn`): same layer structure, same attribute names, hence the same
`state_dict` keys, so the shipped checkpoints (ckpts_*/best.pth.tar) load with `load_state_dict` unchanged.
The per-node dense layers are stock torch.nn; the graph convolution, graph build and MET reduction are this
package's operators.  `graph='dynamic'` rebuilds a kNN graph in the current embedding before every convolution
(the alternative kept at graph_met_network.py:63); `graph='static'` convolves over the `edge_index` argument
(graph_met_network.py:65, the active line).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from . import dense
from .conv import DynamicEdgeConv, EdgeConv
from .scatter import met_loss, met_loss_from_weights, met_reduce

PDG_CLASSES = (1, 2, 11, 13, 22, 130, 211)  # graph_met_network.py:45


def _run(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Apply a Sequential, routing its Linear members through dense.linear (same forward, HIP weight-grad kernel)."""
    for layer in seq:
        x = dense.linear(x, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(x)
    return x


class GraphMETNetwork(nn.Module):
    def __init__(self, continuous_dim: int, cat_dim: int, output_dim: int = 1, hidden_dim: int = 32,
                 conv_depth: int = 1, graph: str = "static", k: int = 16, edge_dtype=None):
        super().__init__()
        if graph not in ("static", "dynamic"):
            raise ValueError("graph must be 'static' or 'dynamic'")
        q, h = hidden_dim // 4, hidden_dim // 2
        self.graph, self.k = graph, k
        self.fused_encoder = os.environ.get("DMET_FUSED_ENCODER", "1") != "0"
        self.embed_charge = nn.Embedding(3, q)
        self.embed_pdgid = nn.Embedding(len(PDG_CLASSES), q)
        self.embed_pv = nn.Embedding(8, q)
        self.embed_continuous = nn.Sequential(nn.Linear(continuous_dim, h), nn.ELU())
        self.embed_categorical = nn.Sequential(nn.Linear(3 * q, h), nn.ELU())
        self.encode_all = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.ELU())
        self.bn_all = nn.BatchNorm1d(hidden_dim)
        self.conv_continuous = nn.ModuleList()
        for _ in range(conv_depth):
            message = nn.Sequential(nn.Linear(2 * hidden_dim, hidden_dim))
            conv = DynamicEdgeConv(nn=message, k=k) if graph == "dynamic" else EdgeConv(nn=message).jittable()
            conv.compute_dtype = edge_dtype   # torch.bfloat16 -> bf16-MFMA edge MLP (BASELINE configs[2])
            self.conv_continuous.append(nn.ModuleList([conv, nn.BatchNorm1d(hidden_dim)]))
        self.output = nn.Sequential(nn.Linear(hidden_dim, h), nn.ELU(), nn.Linear(h, output_dim))
        self.pdgs = list(PDG_CLASSES)

    def _fused_encoder_ok(self, x_cont: torch.Tensor, x_cat: torch.Tensor) -> bool:
        return (self.fused_encoder and x_cont.dtype == torch.float32 and x_cont.shape[1] == 8
                and x_cat.shape[1] == 3 and self.encode_all[0].weight.shape == (32, 32)
                and list(self.pdgs) == list(PDG_CLASSES) and not x_cont.requires_grad)

    def _next_build(self, layer: int, batch, graph=None):
        """The prebuild hook of convolution `layer`: a DynamicEdgeConv's kNN build, or a static EdgeConv's node-level dense
        layer, can carry the BatchNorm transform that produces its input (dense.batch_norm(..., next_build=...)); None
        when there is no such layer or it cannot."""
        if layer >= len(self.conv_continuous):
            return None
        hook = getattr(self.conv_continuous[layer][0], "prebuild_hook", None)
        if hook is None:
            return None
        return hook(batch) if self.graph == "dynamic" else hook(batch, graph=graph)

    def embed(self, x_cont: torch.Tensor, x_cat: torch.Tensor, batch=None, fuse_next: bool = False, graph=None) -> torch.Tensor:
        """Per-node encoder (graph_met_network.py:48-58): columns of x_cat are (pdgId, charge, fromPV).
        The standard shape (8 continuous columns, hidden_dim 32) runs as one HIP kernel each way (csrc/encoder.hip);
        anything else takes the layer-by-layer route below."""
        if self._fused_encoder_ok(x_cont, x_cat):
            lc, lk, la = self.embed_continuous[0], self.embed_categorical[0], self.encode_all[0]
            return dense.encode_bn(x_cont, x_cat, self.bn_all, self._next_build(0, batch, graph) if fuse_next else None,
                                   lc.weight, lc.bias, lk.weight, lk.bias, la.weight, la.bias,
                                   self.embed_charge.weight, self.embed_pdgid.weight, self.embed_pv.weight)
        if x_cat.is_floating_point():      # split_features(x, lazy_cat=True) on the layer-by-layer route
            x_cat = x_cat.long()
        e_cont = _run(self.embed_continuous, x_cont)
        e_chrg = dense.embedding(x_cat[:, 1] + 1, self.embed_charge.weight)
        e_pv = dense.embedding(x_cat[:, 2], self.embed_pv.weight)
        pdg = x_cat[:, 0].abs()
        for cls, val in enumerate(self.pdgs):  # sequential remap, kept sequential for unexpected ids
            pdg = torch.where(pdg == val, torch.full_like(pdg, cls), pdg)
        e_pdg = dense.embedding(pdg, self.embed_pdgid.weight)
        e_cat = _run(self.embed_categorical, torch.cat([e_chrg, e_pdg, e_pv], dim=1))
        return dense.batch_norm(_run(self.encode_all, torch.cat([e_cat, e_cont], dim=1)), self.bn_all)

    def _fused_head_ok(self, emb: torch.Tensor) -> bool:
        l1, l2 = self.output[0], self.output[2]
        return (self.fused_encoder and emb.dtype == torch.float32 and emb.shape[1] == 32 and l1.bias is not None
                and l2.bias is not None and tuple(l1.weight.shape) == (16, 32) and tuple(l2.weight.shape) == (1, 16))

    def forward(self, x_cont, x_cat, edge_index, batch, apply_sigmoid: bool = False):
        """Per-node logit (graph_met_network.py:60-69); apply_sigmoid=True returns sigmoid(logit) instead (what
        Net does), which lets the standard head shape run as one HIP kernel each way (csrc/head.hip)."""
        emb = self.embed(x_cont, x_cat, batch, fuse_next=True, graph=edge_index)
        for layer, (conv, norm) in enumerate(self.conv_continuous):
            # res is emb routed through the conv's autograd node: both gradients of emb meet in its backward
            msg, res = conv.forward_with_residual_input(emb, batch if self.graph == "dynamic" else edge_index)
            # emb + norm(msg) in one streaming pass -- inside the next layer's graph build when that is a kNN build
            nb = self._next_build(layer + 1, batch, edge_index)
            if nb is None and layer + 1 == len(self.conv_continuous) and apply_sigmoid and self._fused_head_ok(res):
                l1, l2 = self.output[0], self.output[2]     # last block: the transform rides in the head's forward launch
                nb = dense.head_prebuild_hook(l1.weight, l1.bias, l2.weight, l2.bias)
            emb = dense.batch_norm(msg, norm, residual=res, next_build=nb)
        if apply_sigmoid and self._fused_head_ok(emb):
            l1, l2 = self.output[0], self.output[2]
            return dense.head(emb, l1.weight, l1.bias, l2.weight, l2.bias)
        out = _run(self.output, emb).squeeze(-1)
        return torch.sigmoid(out) if apply_sigmoid else out


class Net(nn.Module):
    """net.py:38-47: GraphMETNetwork(output_dim=1, hidden_dim=32, conv_depth=2) followed by a sigmoid."""

    def __init__(self, continuous_dim: int, categorical_dim: int, graph: str = "static", k: int = 16, edge_dtype=None):
        super().__init__()
        self.graphnet = GraphMETNetwork(continuous_dim, categorical_dim, output_dim=1, hidden_dim=32,
                                        conv_depth=2, graph=graph, k=k, edge_dtype=edge_dtype)

    def forward(self, x_cont, x_cat, edge_index, batch):
        return self.graphnet(x_cont, x_cat, edge_index, batch, apply_sigmoid=True)


def loss_fn(weights: torch.Tensor, prediction: torch.Tensor, truth: torch.Tensor, batch: torch.Tensor,
            ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """net.py:49-62 with the two scatter_add calls fused into one MET reduction:
    0.5 * mean_b((METx + true_px)^2 + (METy + true_py)^2)."""
    if weights.dtype == torch.float32 and prediction.dtype == torch.float32 and truth.dtype == torch.float32:
        return met_loss_from_weights(weights, prediction, truth, batch, ptr=ptr)
    met = met_reduce(weights, prediction, batch, ptr=ptr, num_events=truth.shape[0])
    return met_loss(met, truth)


def split_features(x: torch.Tensor, lazy_cat: bool = False):
    """train.py:42-46: continuous columns 0..7 (puppi included), categorical columns 8..10 as int64.
    lazy_cat=True hands the categorical columns over as the float view x[:, 8:]: the fused encoder kernel converts them
    itself (same truncation as `.long()`), which saves the conversion kernel and its [N,3] int64 tensor per step; any
    other consumer inside this package converts on demand."""
    return x[:, :8], (x[:, 8:] if lazy_cat else x[:, 8:].long())
