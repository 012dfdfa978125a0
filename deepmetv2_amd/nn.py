"""`torch.nn` with the three per-node layer types of the graph-MET model on this package's kernels.

A FOURTH import line for the maintainer of /root/reference/model/graph_met_network.py (`import torch.nn as nn`, line 3
there) -- optional, the three operator imports of INTEGRATION.md work without it:

    import deepmetv2_amd.nn as nn

Every name of `torch.nn` is re-exported unchanged except `Linear`, `Embedding` and `BatchNorm1d`, which are SUBCLASSES of
torch's (same constructor, same parameters / buffers, same `state_dict` keys, `isinstance(m, torch.nn.Linear)` holds):

  * Linear      forward = the library GEMM; backward's weight gradient (K = 288 000 nodes, a 32 x 32 result: rocBLAS picks a
                32x32x256 tile and takes ~550 us per layer) through the fp32-MFMA reduction of csrc/dense.hip (~15 us);
  * Embedding   backward through a one-hot x gradient product (csrc/dense.hip) instead of torch's sort-based
                `embedding_dense_backward`, which needs ~1.7 ms per table for 288 000 indices into 3-8 rows (three tables
                = half of the drop-in training step, profiles/r03_stock_kernel_stats.csv);
  * BatchNorm1d one statistics pass + one transform pass each way (csrc/norm.hip), running statistics as torch's.

Inputs the kernels are not built for (CPU tensors, other dtypes / ranks, exotic constructor options) fall through to the
torch implementation of the parent class.  `accelerate(model)` applies the same swap to an existing model in place and,
when the model has the graph-MET structure, returns this package's fused `model.Net` sharing its parameters.
"""
from __future__ import annotations

import torch
import torch.nn as _nn

from . import dense as _dense

# re-export torch.nn
globals().update({_k: getattr(_nn, _k) for _k in dir(_nn) if not _k.startswith("_")})


def _plain_2d(x: torch.Tensor) -> bool:
    return torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32


class Linear(_nn.Linear):
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if _plain_2d(input) and self.weight.dtype == torch.float32:
            return _dense.linear(input, self.weight, self.bias)
        return super().forward(input)


class Embedding(_nn.Embedding):
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if (torch.is_tensor(input) and input.is_cuda and input.dim() == 1 and input.dtype == torch.int64
                and self.padding_idx is None and self.max_norm is None and not self.sparse
                and not self.scale_grad_by_freq and self.weight.dtype == torch.float32):
            return _dense.embedding(input, self.weight)
        return super().forward(input)


class BatchNorm1d(_nn.BatchNorm1d):
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if _plain_2d(input):
            return _dense.batch_norm(input, self)      # falls back to the module's own forward where it must
        return super().forward(input)

    def _torch_forward(self, input):
        return super().forward(input)


_SWAP = {_nn.Linear: Linear, _nn.Embedding: Embedding, _nn.BatchNorm1d: BatchNorm1d}

_GRAPHMET_CHILDREN = ("embed_charge", "embed_pdgid", "embed_pv", "embed_continuous", "embed_categorical", "encode_all",
                      "bn_all", "conv_continuous", "output")


def _graphmet_shape(model: _nn.Module):
    """(continuous_dim, conv_depth, k or None) when `model` is wired like the reference's Net -- a `graphnet` child with
    the attribute names and shapes of /root/reference/model/graph_met_network.py:11-44 at hidden width 32 -- else None."""
    g = getattr(model, "graphnet", None)
    if g is None or any(not hasattr(g, name) for name in _GRAPHMET_CHILDREN):
        return None
    try:
        if g.encode_all[0].weight.shape != (32, 32) or g.output[0].weight.shape != (16, 32) or g.output[2].weight.shape != (1, 16):
            return None
        if g.embed_charge.weight.shape != (3, 8) or g.embed_pdgid.weight.shape != (7, 8) or g.embed_pv.weight.shape != (8, 8):
            return None
        k = None
        for blk in g.conv_continuous:
            conv, norm = blk[0], blk[1]
            if conv.nn[0].weight.shape != (32, 64) or norm.num_features != 32 or getattr(conv, "aggr", "max") != "max":
                return None
            k = getattr(conv, "k", k)
        return int(g.embed_continuous[0].weight.shape[1]), len(g.conv_continuous), k
    except (AttributeError, IndexError, TypeError):
        return None


def _share_state(dst: _nn.Module, src: _nn.Module) -> None:
    """dst's parameters and buffers BECOME src's objects (same names): an optimizer built on either sees both, a
    checkpoint of either loads into the other."""
    src_mods = dict(src.named_modules())
    for name, mod in dst.named_modules():
        peer = src_mods.get(name)
        if peer is None:
            continue
        for pname in list(mod._parameters):
            if pname in peer._parameters:
                mod._parameters[pname] = peer._parameters[pname]
        for bname in list(mod._buffers):
            if bname in peer._buffers:
                mod._buffers[bname] = peer._buffers[bname]


def accelerate(model: _nn.Module, graph: str = "auto", k: int = 16, fuse: bool = True) -> _nn.Module:
    """One line in the training script after the model is built and moved to the device (cf. /root/reference/train.py:73):

        model = deepmetv2_amd.accelerate(model)

    * every `torch.nn.Linear` / `Embedding` / `BatchNorm1d` inside `model` (exactly these classes, not user subclasses)
      becomes the subclass above, in place: same parameters, same state_dict, HIP kernels for the slow halves;
    * `fuse=True` and a model wired like the reference's `Net` (hidden width 32; attribute names as in its checkpoints):
      the return value is this package's `model.Net` -- fused encoder / head kernels, BatchNorm transforms riding in the
      kNN builds -- SHARING the parameters and buffers of `model`; same call signature
      `(x_cont, x_cat, edge_index, batch)`.  graph: 'dynamic' = rebuild a kNN graph (k) in the embedding before every
      convolution (graph_met_network.py:63, what DynamicEdgeConv blocks mean), 'static' = convolve over the `edge_index`
      argument (:65); 'auto' = 'dynamic' if the convolutions are DynamicEdgeConv, else 'static'.
    Anything else is returned as it came (with the layer swap applied)."""
    for mod in model.modules():
        swap = _SWAP.get(type(mod))
        if swap is not None:
            mod.__class__ = swap
    shape = _graphmet_shape(model) if fuse else None
    if shape is None:
        return model
    cont_dim, depth, conv_k = shape
    from .conv import DynamicEdgeConv
    from .model import Net
    if graph == "auto":
        graph = "dynamic" if all(isinstance(b[0], DynamicEdgeConv) for b in model.graphnet.conv_continuous) else "static"
    if graph not in ("dynamic", "static"):
        raise ValueError(f"graph must be 'auto', 'dynamic' or 'static', got {graph!r}")
    if depth != 2:
        return model
    was_training = model.training
    p0 = next(model.parameters())
    fused = Net(cont_dim, 3, graph=graph, k=int(conv_k or k)).to(device=p0.device)
    _share_state(fused, model)
    fused.train(was_training)
    return fused
