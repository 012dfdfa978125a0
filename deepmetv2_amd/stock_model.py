"""The drop-in view: the graph-MET model as a maintainer of the reference has it after changing ONLY the three import
lines of INTEGRATION.md -- every layer stock `torch.nn`, the graph operators taken from an `ops` namespace that offers
the third-party names (`EdgeConv`, `DynamicEdgeConv`, `knn_graph`, `scatter_add`).  With `ops = deepmetv2_amd` that is
this package's public surface and nothing else: no fused encoder / head kernels, no BatchNorm riders, no
`forward_with_residual_input`, no flat parameter buffer; `torch.optim.AdamW` runs on `model.parameters()`.

This is this repo's own restatement of the wiring (what it restates: /root/reference/model/graph_met_network.py:11-69,
model/net.py:38-62, train.py:40-52), written for `bench.py --model stock-*` and `tests/test_gpu_stock.py`: it exists to
TIME and TEST what the reference's call shapes get from the operators, next to the headline number of this repo's own
fused `model.Net`.  The convolution block comes in the three spellings the reference knows:

  "knn_graph" (graph_met_network.py:63, the commented dynamic line; conv = EdgeConv(nn).jittable()):
        h + norm(conv(h, knn_graph(h, k=k, batch=batch, loop=True)))
  "dynamic"   (PyG's DynamicEdgeConv(nn, k), what :63 amounts to):
        h + norm(conv(h, batch))
  "static"    (graph_met_network.py:65, the active line): h + norm(conv(h, edge_index)) on the caller's radius graph

Module attribute names equal the reference's, so its checkpoints and `model.Net`'s state_dicts load unchanged.
"""
from __future__ import annotations

import torch
from torch import nn

PDG_CODES = (1, 2, 11, 13, 22, 130, 211)      # class index = position; other |pdgId| values pass through unchanged
VARIANTS = ("knn_graph", "dynamic", "static")


def _mlp(n_in: int, n_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(n_in, n_out), nn.ELU())


class StockGraphMETNetwork(nn.Module):
    def __init__(self, ops, continuous_dim: int, cat_dim: int, output_dim: int = 1, hidden_dim: int = 32,
                 conv_depth: int = 1, variant: str = "knn_graph", k: int = 16):
        super().__init__()
        if variant not in VARIANTS:
            raise ValueError(f"variant must be one of {VARIANTS}, got {variant!r}")
        self.ops, self.variant, self.k = ops, variant, int(k)
        width, half, quarter = hidden_dim, hidden_dim // 2, hidden_dim // 4
        for name, rows in (("embed_charge", 3), ("embed_pdgid", len(PDG_CODES)), ("embed_pv", 8)):
            setattr(self, name, nn.Embedding(rows, quarter))
        self.embed_continuous = _mlp(continuous_dim, half)
        self.embed_categorical = _mlp(3 * quarter, half)
        self.encode_all = _mlp(width, width)
        self.bn_all = nn.BatchNorm1d(width)
        blocks = []
        for _ in range(conv_depth):
            edge_nn = nn.Sequential(nn.Linear(2 * width, width))
            if variant == "dynamic":
                conv = ops.DynamicEdgeConv(nn=edge_nn, k=self.k)
            else:
                conv = ops.EdgeConv(nn=edge_nn).jittable()
            blocks.append(nn.ModuleList([conv, nn.BatchNorm1d(width)]))
        self.conv_continuous = nn.ModuleList(blocks)
        self.output = nn.Sequential(nn.Linear(width, half), nn.ELU(), nn.Linear(half, output_dim))

    def _pdg_class(self, pdg_id: torch.Tensor) -> torch.Tensor:
        cls = pdg_id.abs()
        for code, value in enumerate(PDG_CODES):        # one after the other, as the reference does
            cls = torch.where(cls == value, torch.full_like(cls, code), cls)
        return cls

    def encode(self, x_cont: torch.Tensor, x_cat: torch.Tensor) -> torch.Tensor:
        pdg_id, charge, from_pv = x_cat.unbind(dim=1)
        tokens = torch.cat([self.embed_charge(charge + 1), self.embed_pdgid(self._pdg_class(pdg_id)),
                            self.embed_pv(from_pv)], dim=1)
        joint = torch.cat([self.embed_categorical(tokens), self.embed_continuous(x_cont)], dim=1)
        return self.bn_all(self.encode_all(joint))

    def forward(self, x_cont, x_cat, edge_index, batch):
        ops, h = self.ops, self.encode(x_cont, x_cat)
        for conv, norm in self.conv_continuous:
            if self.variant == "knn_graph":
                h = h + norm(conv(h, ops.knn_graph(h, k=self.k, batch=batch, loop=True)))
            elif self.variant == "dynamic":
                h = h + norm(conv(h, batch))
            else:
                h = h + norm(conv(h, edge_index))
        return self.output(h).squeeze(-1)


class StockNet(nn.Module):
    """Hidden width 32, two convolution blocks, sigmoid on the per-node logit."""

    def __init__(self, ops, continuous_dim: int, categorical_dim: int, variant: str = "knn_graph", k: int = 16):
        super().__init__()
        self.graphnet = StockGraphMETNetwork(ops, continuous_dim, categorical_dim, output_dim=1, hidden_dim=32,
                                             conv_depth=2, variant=variant, k=k)

    def forward(self, x_cont, x_cat, edge_index, batch):
        return self.graphnet(x_cont, x_cat, edge_index, batch).sigmoid()


def stock_loss_fn(ops, weights, prediction, truth, batch):
    """Half the mean squared MET residual, the per-event sums through TWO scatter_add calls as at model/net.py:55-56."""
    met_x = ops.scatter_add(weights * prediction[:, 0], batch)
    met_y = ops.scatter_add(weights * prediction[:, 1], batch)
    residual_sq = (met_x + truth[:, 0]).square() + (met_y + truth[:, 1]).square()
    return 0.5 * residual_sq.mean()


def stock_train_step(ops, model, optimizer, x, y, batch, graph_fn=None):
    """The sequence of train.py:40-52: zero_grad, feature split, (static graph), model, loss, backward, step."""
    optimizer.zero_grad()
    edge_index = None if graph_fn is None else graph_fn(x)
    weights = model(x[:, :8], x[:, 8:].long(), edge_index, batch)
    loss = stock_loss_fn(ops, weights, x, y, batch)
    loss.backward()
    optimizer.step()
    return loss.detach()
