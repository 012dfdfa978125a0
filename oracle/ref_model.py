"""CPU restatement (ORACLE, test infrastructure only) of the reference's model wiring on top of oracle/ref_ops.py.

PARITY UNPINNED for the graph operators (see ref_ops.py).  The wiring follows
/root/reference/model/graph_met_network.py:11-69 and /root/reference/model/net.py:38-47 (cited per step below); in
this container it is additionally checked against the reference's OWN model file imported over oracle-backed
stand-ins of the three missing third-party modules (oracle/gen_golden.py writes that run into tests/golden/,
tests/test_oracle.py compares).  Module attribute names equal the reference's so that state_dicts interchange.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ref_ops

_PDG_TABLE = (1, 2, 11, 13, 22, 130, 211)      # graph_met_network.py:45


class RefEdgeConv(nn.Module):
    """PyG EdgeConv shell: holds the caller's `nn` under the attribute name `.nn`, owns nothing else."""

    def __init__(self, nn, aggr: str = "max", flow: str = "source_to_target", **kwargs):
        super().__init__()
        self.nn, self.aggr, self.flow = nn, aggr, flow

    def jittable(self, *a, **k):
        return self

    def forward(self, x, edge_index):
        return ref_ops.edge_conv(x, edge_index, self.nn, self.aggr, self.flow)


class RefDynamicEdgeConv(RefEdgeConv):
    def __init__(self, nn, k: int, aggr: str = "max", **kwargs):
        super().__init__(nn, aggr)
        self.k = k

    def forward(self, x, batch=None):
        return ref_ops.dynamic_edge_conv(x, batch, self.nn, self.k, self.aggr)


def _dense(n_in: int, n_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(n_in, n_out), nn.ELU())


class RefGraphMETNetwork(nn.Module):
    def __init__(self, continuous_dim, cat_dim, output_dim=1, hidden_dim=32, conv_depth=1, graph="static", k=16):
        super().__init__()
        H = hidden_dim
        self.graph = graph
        # three categorical embeddings of width H/4 (:15-17): charge in {-1,0,1}+1, 7 pdg classes, fromPV < 8
        self.embed_charge, self.embed_pdgid, self.embed_pv = (nn.Embedding(n, H // 4) for n in (3, len(_PDG_TABLE), 8))
        self.embed_continuous = _dense(continuous_dim, H // 2)          # :19-22
        self.embed_categorical = _dense(3 * H // 4, H // 2)             # :24-27
        self.encode_all = _dense(H, H)                                  # :29-31
        self.bn_all = nn.BatchNorm1d(H)                                 # :32
        blocks = []
        for _ in range(conv_depth):                                     # :34-39: EdgeConv(Linear(2H,H)) then BN
            message = nn.Sequential(nn.Linear(2 * H, H))
            conv = RefDynamicEdgeConv(message, k) if graph == "dynamic" else RefEdgeConv(message)
            blocks.append(nn.ModuleList([conv, nn.BatchNorm1d(H)]))
        self.conv_continuous = nn.ModuleList(blocks)
        self.output = nn.Sequential(nn.Linear(H, H // 2), nn.ELU(), nn.Linear(H // 2, output_dim))   # :41-44

    def node_embedding(self, x_cont, x_cat):
        pdg_col, charge_col, pv_col = x_cat[:, 0], x_cat[:, 1], x_cat[:, 2]
        cls = pdg_col.abs()                                             # :52
        for code, value in enumerate(_PDG_TABLE):                       # :53-54, sequential on purpose
            cls = torch.where(cls == value, torch.full_like(cls, code), cls)
        cat_vec = torch.cat([self.embed_charge(charge_col + 1),         # :49
                             self.embed_pdgid(cls),                     # :55
                             self.embed_pv(pv_col)], dim=1)             # :50 ; order charge,pdg,pv per :57
        joint = torch.cat([self.embed_categorical(cat_vec), self.embed_continuous(x_cont)], dim=1)   # :48,:57,:58
        return self.bn_all(self.encode_all(joint))                      # :58

    def forward(self, x_cont, x_cat, edge_index, batch):
        h = self.node_embedding(x_cont, x_cat)
        graph_arg = batch if self.graph == "dynamic" else edge_index    # :63 (dynamic kNN) vs :65 (static)
        for conv, norm in self.conv_continuous:                         # :61
            h = h + norm(conv(h, graph_arg))                            # residual around BN(EdgeConv)
        return self.output(h).squeeze(-1)                               # :67,:69


class RefNet(nn.Module):
    """net.py:38-47: hidden 32, two convolutions, sigmoid on the per-node logit."""

    def __init__(self, continuous_dim, categorical_dim, graph="static", k=16):
        super().__init__()
        self.graphnet = RefGraphMETNetwork(continuous_dim, categorical_dim, 1, 32, 2, graph=graph, k=k)

    def forward(self, x_cont, x_cat, edge_index, batch):
        return self.graphnet(x_cont, x_cat, edge_index, batch).sigmoid()
