#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container; commits are data only).

    python oracle/gen_golden.py [--reference /root/reference]

What pins what
  * The reference (DeepMETv2) ships no tests and no golden vectors, and its operators' arithmetic lives in
    torch_cluster / torch_scatter / torch_geometric, absent from the image: PARITY UNPINNED for the operators.
    G1/G2/G3/G5 below are therefore outputs of THIS repo's CPU oracle (oracle/dmet_oracle.c, oracle/ref_ops.py); they
    freeze the restated rules R1-R6 so that later edits of oracle or kernels cannot drift silently.
  * G4 is stronger: it runs the REFERENCE'S OWN model code (model/net.py + model/graph_met_network.py, imported from
    --reference) with the shipped checkpoint ckpts_dytt/best.pth.tar (torch.load(weights_only=True)) on seeded
    synthetic events.  Only the three missing third-party modules are replaced by oracle-backed stand-ins
    (radius_graph/knn_graph -> ref_ops, EdgeConv -> ref_model.RefEdgeConv, scatter_add -> ref_ops.scatter_add).
    The wiring, embeddings, BatchNorm, residuals, sigmoid and the loss are the reference's own lines.
    Nothing of the reference's source text is stored: the fixture holds inputs and outputs only.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepmetv2_amd import synth  # noqa: E402  (pure-torch CPU generator, no HIP involved)
from oracle import ref_model, ref_ops  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def _ragged(sizes, D, seed, dup=False):
    g = torch.Generator().manual_seed(seed)
    N = sum(sizes)
    x = torch.randn(N, D, generator=g)
    if dup and N >= 8:
        x[N // 3] = x[1]
        x[N // 2] = x[1]
        x[-(N // 4):] = torch.round(x[-(N // 4):])
    counts = torch.tensor(sizes)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    batch = torch.repeat_interleave(torch.arange(len(sizes)), counts)
    return x, batch, ptr


def g1_config1(ckpt_state):
    """BASELINE configs[0]: 1 event, 256 nodes, H=32, k=8, one EdgeConv(Linear 64->32); seeded and trained weights."""
    x, batch, ptr = _ragged([256], 32, seed=1)
    nbr, dist = ref_ops.knn_table(x, ptr, 8)
    out = {}
    torch.manual_seed(11)
    lin = torch.nn.Linear(64, 32)
    sets = {"seeded": (lin.weight.detach().clone(), lin.bias.detach().clone())}
    if ckpt_state is not None:
        sets["trained"] = (ckpt_state["graphnet.conv_continuous.0.0.nn.0.weight"],
                           ckpt_state["graphnet.conv_continuous.0.0.nn.0.bias"])
    ei = ref_ops.knn_graph(x, 8, None, loop=True)
    for tag, (W, b) in sets.items():
        with torch.no_grad():
            lin.weight.copy_(W); lin.bias.copy_(b)
        o, arg = ref_ops.edge_conv(x, ei, lin, return_arg=True)
        out[f"W_{tag}"], out[f"b_{tag}"], out[f"out_{tag}"] = W, b, o
        out[f"argslot_{tag}"] = (arg % 8).to(torch.uint8)  # edges are grouped 8 per node: slot = edge % k
    _save("g1_config1.npz", x=x, ptr=ptr, nbr=nbr, dist=dist, **out)


def g2_ties():
    """R2/R3 stress: duplicated rows, lattice points (exact ties), n < k, single-node and EMPTY events."""
    x, batch, ptr = _ragged([1, 3, 0, 17, 129, 64, 2], 8, seed=2, dup=True)
    nbr, dist = ref_ops.knn_table(x, ptr, 16)
    nbr_py = ref_ops.knn_table_pyloops(x, ptr, 16)
    assert torch.equal(nbr, nbr_py), "C oracle and the independent Python-loop restatement disagree"
    ei_noloop = ref_ops.knn_graph(x, 4, batch, loop=False)
    _save("g2_ties.npz", x=x, ptr=ptr, batch=batch, nbr16=nbr, dist16=dist, ei_k4_noloop=ei_noloop)


def g3_ragged():
    """config 5 scaled down: ragged 3-event batch 50/450/800, k=16, DynamicEdgeConv output with seeded weights."""
    x, batch, ptr = _ragged([50, 450, 800], 32, seed=3)
    nbr, _ = ref_ops.knn_table(x, ptr, 16)
    torch.manual_seed(12)
    lin = torch.nn.Sequential(torch.nn.Linear(64, 32))
    out = ref_ops.dynamic_edge_conv(x, batch, lin, 16)
    _save("g3_ragged.npz", x=x, ptr=ptr, batch=batch, nbr=nbr, W=lin[0].weight, b=lin[0].bias, out=out)


def g5_irregular():
    """DRN call shape: loop=False kNN, symmetrised + shuffled edge list, aggr in {max, add}, 2-layer nn."""
    x, batch, ptr = _ragged([60, 5, 90], 16, seed=21)
    torch.manual_seed(13)
    nn_ = torch.nn.Sequential(torch.nn.Linear(32, 24), torch.nn.ELU(), torch.nn.Linear(24, 16), torch.nn.ELU())
    ei = ref_ops.knn_graph(x, 4, batch, loop=False)
    ei = torch.cat([ei, ei.flip(0)], dim=1)
    ei = ei[:, torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))]
    o_max = ref_ops.edge_conv(x, ei, nn_, "max")
    o_add = ref_ops.edge_conv(x, ei, nn_, "add")
    sd = {k.replace(".", "_"): v for k, v in nn_.state_dict().items()}
    _save("g5_irregular.npz", x=x, edge_index=ei, out_max=o_max, out_add=o_add, **sd)


def g6_radius():
    g = torch.Generator().manual_seed(3)
    sizes = [300, 5, 1000]
    N = sum(sizes)
    etaphi = torch.stack([(torch.rand(N, generator=g) - 0.5) * 6, (torch.rand(N, generator=g) - 0.5) * 6.28], 1)
    batch = torch.repeat_interleave(torch.arange(3), torch.tensor(sizes))
    ei = ref_ops.radius_graph(etaphi, 0.4, batch, loop=True, max_num_neighbors=255)
    ei12 = ref_ops.radius_graph(etaphi, 0.4, batch, loop=False, max_num_neighbors=12)
    _save("g6_radius.npz", etaphi=etaphi, batch=batch, ei_r04_loop_255=ei.to(torch.int32),
          ei_r04_noloop_12=ei12.to(torch.int32))


def _install_standins():
    """Stand-ins for the three third-party modules the reference imports (absent from the image)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Unused:  # names imported by the reference but never called on this path
        def __init__(self, *a, **k):
            raise NotImplementedError

    tg = mod("torch_geometric")
    tg.nn = mod("torch_geometric.nn", EdgeConv=ref_model.RefEdgeConv, NNConv=_Unused, graclus=_Unused, max_pool=_Unused,
                max_pool_x=_Unused, global_mean_pool=_Unused, global_max_pool=_Unused, global_add_pool=_Unused)
    tg.nn.conv = mod("torch_geometric.nn.conv", EdgeConv=ref_model.RefEdgeConv, GraphConv=_Unused, GCNConv=_Unused)
    tg.nn.pool = mod("torch_geometric.nn.pool")
    tg.nn.pool.edge_pool = mod("torch_geometric.nn.pool.edge_pool", EdgePooling=_Unused)
    tg.transforms = mod("torch_geometric.transforms", Cartesian=lambda **k: None)
    tg.utils = mod("torch_geometric.utils", normalized_cut=_Unused, remove_self_loops=_Unused, to_undirected=_Unused)
    tg.utils.undirected = mod("torch_geometric.utils.undirected", to_undirected=_Unused)
    mod("torch_cluster", knn_graph=ref_ops.knn_graph, radius_graph=ref_ops.radius_graph)
    mod("torch_scatter", scatter_add=ref_ops.scatter_add)


def g4_reference_model(reference: str):
    """The reference's own Net + loss_fn on seeded events with its shipped trained weights."""
    ckpt = torch.load(os.path.join(reference, "ckpts_dytt", "best.pth.tar"), map_location="cpu", weights_only=True)
    _install_standins()
    sys.path.insert(0, reference)
    try:
        net = importlib.import_module("model.net")
    finally:
        sys.path.remove(reference)
    x, y, batch, ptr = synth.make_events([300, 40, 500], seed=4)
    x_cont, x_cat = x[:, :8], x[:, 8:].long()                     # train.py:42-44
    phi = torch.atan2(x[:, 1], x[:, 0])                           # train.py:45
    etaphi = torch.cat([x[:, 3][:, None], phi[:, None]], dim=1)   # train.py:46
    edge_index = ref_ops.radius_graph(etaphi, r=0.4, batch=batch, loop=True, max_num_neighbors=255)   # train.py:48
    out = {}
    for mode in ("eval", "train"):
        model = net.Net(8, 3)
        model.load_state_dict(ckpt["state_dict"])
        getattr(model, mode)()
        w = model(x_cont, x_cat, edge_index, batch)               # train.py:49
        loss = net.loss_fn(w, x, y, batch)                        # train.py:50
        loss.backward()
        out[f"weights_{mode}"] = w
        out[f"loss_{mode}"] = loss
        out[f"grad_conv0_weight_{mode}"] = model.graphnet.conv_continuous[0][0].nn[0].weight.grad
        out[f"grad_output2_weight_{mode}"] = model.graphnet.output[2].weight.grad
    state = {k.replace(".", "__"): v for k, v in ckpt["state_dict"].items()}
    _save("g4_reference_model.npz", x=x, y=y, batch=batch, ptr=ptr, edge_index=edge_index.to(torch.int32), **out)
    _save("g4_checkpoint_dytt_best.npz", **state)
    return ckpt["state_dict"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    torch.set_num_threads(8)
    state = None
    if os.path.isdir(args.reference):
        state = g4_reference_model(args.reference)
    else:
        print(f"{args.reference} not found: G4 (reference model run) skipped")
    g1_config1(state)
    g2_ties()
    g3_ragged()
    g5_irregular()
    g6_radius()


if __name__ == "__main__":
    main()
