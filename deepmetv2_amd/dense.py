"""Per-node dense layers with deterministic HIP weight-gradient kernels (first piece of SURVEY.md row N3).

`linear` / `embedding` compute exactly what torch.nn.functional.linear / embedding compute in forward (library GEMM /
index_select -- plumbing); only the backward differs: the tall-skinny weight gradients go through csrc/dense.hip
instead of a library GEMM / a sort-based scatter.  Modules keep their torch.nn.Linear / Embedding parameters (same
state_dict); these functions are called with those parameters.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import _native


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = _native.xty(g, x) if ctx.needs_input_grad[1] else None
        gb = g.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, index, weight):
        ctx.save_for_backward(index)
        ctx.rows = weight.shape[0]
        return weight.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return None, _native.onehot_xty(index, g.contiguous(), ctx.rows)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ weight^T + bias for x[N,in] (in,out <= 64)."""
    if x.dim() != 2 or weight.shape[0] > 64 or weight.shape[1] > 64 or x.dtype != torch.float32:
        return torch.nn.functional.linear(x, weight, bias)
    return _Linear.apply(x, weight, bias)


def embedding(index: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """weight[index] for a small table (rows, width <= 64)."""
    if weight.shape[0] > 64 or weight.shape[1] > 64 or index.dim() != 1:
        return torch.nn.functional.embedding(index, weight)
    return _Embedding.apply(index, weight)


class _Encode(torch.autograd.Function):
    """The whole per-node encoder (graph_met_network.py:48-58 before bn_all) as one HIP kernel each way
    (csrc/encoder.hip).  Inputs carry no gradient; the nine parameter gradients come out of one backward kernel."""

    @staticmethod
    def forward(ctx, x_cont, x_cat, *params):
        h = _native.encode_fwd(x_cont, x_cat, [p.detach() for p in params])
        ctx.save_for_backward(x_cont, x_cat, h, *params)
        return h

    @staticmethod
    def backward(ctx, g_h):
        x_cont, x_cat, h, *params = ctx.saved_tensors
        grads = _native.encode_bwd(x_cont, x_cat, [p.detach() for p in params], h, g_h.contiguous())
        return (None, None, *grads)


def encode(x_cont: torch.Tensor, x_cat: torch.Tensor, *params: torch.Tensor) -> torch.Tensor:
    """h[N,32] = ELU(Wa [ELU(Wk [Echg[chg+1] | Epdg[remap |pdg|] | Epv[pv]] + bk) | ELU(Wc x_cont + bc)] + ba);
    params = (Wc, bc, Wk, bk, Wa, ba, Echg, Epdg, Epv) in torch layouts."""
    return _Encode.apply(x_cont, x_cat, *params)


class _BatchNorm(torch.autograd.Function):
    """BatchNorm1d over the rows of x[N,H] (+ residual) on the HIP kernels of csrc/norm.hip."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, tracked=None,
                next_build=None):
        y = None
        if next_build is not None and x.is_cuda and x.shape[1] == 32 and (training or running_mean is not None):
            # the transform rides in the prep launch of the NEXT layer's graph build (which consumes y): statistics here
            # (eval mode: the running ones), then one pass that writes y and cuts the build's tile records from it
            # (dmet_bn_knn_local_dense_f32; before the head: dmet_bn_head_fwd_f32)
            if training:
                mean, invstd = _native.bn_stats(x, eps, momentum, running_mean, running_var, tracked)
            else:
                mean, invstd = _native.bn_eval_stats(running_mean, running_var, eps)
            y = next_build(x, residual, weight.detach(), bias.detach(), mean, invstd)
            if y is None:   # that build takes another path after all: the transform alone (the statistics are done),
                y = _native.bn_apply(x, residual, weight.detach(), bias.detach(), mean, invstd)   # bn_fwd's kernel and bits
        if y is None:
            y, mean, invstd = _native.bn_fwd(x, residual, weight.detach(), bias.detach(), eps, momentum,
                                             running_mean, running_var, training, num_batches_tracked=tracked)
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.training = training
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, weight, mean, invstd = ctx.saved_tensors
        g_y = g_y.contiguous()
        if ctx.training:
            g_x, g_w, g_b = _native.bn_bwd(x, g_y, weight.detach(), mean, invstd)
        else:   # running statistics are constants: a per-channel affine map
            scale = weight.detach() * invstd
            g_x = g_y * scale
            g_w = (g_y * (x - mean) * invstd).sum(0)
            g_b = g_y.sum(0)
        return g_x, (g_y if ctx.has_residual else None), g_w, g_b, None, None, None, None, None, None, None


ENC_BN_FUSE = os.environ.get("DMET_ENC_BN_FUSE", "1")


class _EncodeBN(torch.autograd.Function):
    """bn_all(encode(x)) of graph_met_network.py:48-58 in training mode as ONE autograd node: forward = _Encode's and
    _BatchNorm's (the transform inside the next graph build when `next_build` says so); backward = the BatchNorm's column
    sums, then the encoder's backward kernel with the BatchNorm's backward transform applied to the rows as it loads them
    (dmet_encode_bn_bwd_f32) -- h, the BatchNorm's input, is the encoder's own output, which that kernel reads anyway."""

    @staticmethod
    def forward(ctx, x_cont, x_cat, weight, bias, running_mean, running_var, momentum, eps, tracked, next_build, *params):
        h = _native.encode_fwd(x_cont, x_cat, [p.detach() for p in params])
        y = None
        if next_build is not None:
            mean, invstd = _native.bn_stats(h, eps, momentum, running_mean, running_var, tracked)
            y = next_build(h, None, weight.detach(), bias.detach(), mean, invstd)
            if y is None:
                y = _native.bn_apply(h, None, weight.detach(), bias.detach(), mean, invstd)
        if y is None:
            y, mean, invstd = _native.bn_fwd(h, None, weight.detach(), bias.detach(), eps, momentum, running_mean,
                                             running_var, True, num_batches_tracked=tracked)
        ctx.save_for_backward(x_cont, x_cat, h, weight, mean, invstd, *params)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x_cont, x_cat, h, weight, mean, invstd, *params = ctx.saved_tensors
        g_y = g_y.contiguous()
        ps = [p.detach() for p in params]
        res = _native.encode_bn_bwd(x_cont, x_cat, ps, h, g_y, weight.detach(), mean, invstd)
        if res is None:     # the fused backward is not available for these operands: the two separate steps
            g_h, g_w, g_b = _native.bn_bwd(h, g_y, weight.detach(), mean, invstd)
            grads = _native.encode_bwd(x_cont, x_cat, ps, h, g_h)
        else:
            grads, g_w, g_b = res
        return (None, None, g_w, g_b, None, None, None, None, None, None, *grads)


def encode_bn(x_cont: torch.Tensor, x_cat: torch.Tensor, bn: torch.nn.BatchNorm1d, next_build, *params: torch.Tensor):
    """bn(encode(x_cont, x_cat, *params)) -- batch_norm(encode(...), bn, next_build=next_build) with the BatchNorm's
    backward transform inside the encoder's backward kernel when bn is a plain training-mode BatchNorm1d(32) on the GPU
    (DMET_ENC_BN_FUSE=0: always the two nodes)."""
    ok = (ENC_BN_FUSE != "0" and x_cont.is_cuda and bn.training and bn.affine and bn.momentum is not None
          and bn.track_running_stats and bn.num_features == 32 and x_cont.shape[0] > 1)
    if not ok:
        return batch_norm(encode(x_cont, x_cat, *params), bn, next_build=next_build)
    return _EncodeBN.apply(x_cont, x_cat, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.momentum),
                           float(bn.eps), bn.num_batches_tracked, next_build, *params)


def batch_norm(x: torch.Tensor, bn: torch.nn.BatchNorm1d, residual: Optional[torch.Tensor] = None,
               next_build=None) -> torch.Tensor:
    """residual + bn(x) (residual optional) for a torch.nn.BatchNorm1d module `bn` over x[N,H]: same parameters,
    buffers and train/eval semantics as calling the module (momentum=None, no affine or H not a multiple of 4 up to
    64 take the module itself)."""
    H = x.shape[1] if x.dim() == 2 else -1
    ok = (x.dim() == 2 and x.dtype == torch.float32 and H % 4 == 0 and 4 <= H <= 64 and bn.affine
          and bn.momentum is not None and x.shape[0] > 1 and (bn.training or bn.track_running_stats))
    if not ok:
        # (deepmetv2_amd.nn.BatchNorm1d calls this function from its own forward: take torch's forward then)
        own = getattr(bn, "_torch_forward", None)
        y = own(x) if own is not None else bn(x)
        return y if residual is None else residual + y
    training = bn.training or not bn.track_running_stats
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    # the statistics kernel bumps num_batches_tracked (torch's own `add_(1)` is one more launch per layer and step)
    tracked = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
    # next_build (DynamicEdgeConv.prebuild of the layer that consumes the result, training mode): the transform is fused
    # into that layer's graph build
    return _BatchNorm.apply(x, residual, bn.weight, bn.bias, rm, rv, training, float(bn.momentum), float(bn.eps), tracked,
                            next_build)


# (emb, out) of the last dmet_bn_head_fwd_f32 call, until the head that consumes emb asks for it
_HEAD_PREBUILT = [None]
HEAD_FUSE = os.environ.get("DMET_BN_HEAD_FUSE", "1")


def head_prebuild_hook(W1: torch.Tensor, b1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor):
    """A callable for batch_norm(..., next_build=...) when the BatchNorm's result goes straight into the output head
    (model/graph_met_network.py:66-67): the transform is formed inside the head's forward launch, which also writes emb
    (dmet_bn_head_fwd_f32); head(emb, ...) on that emb then returns the result that is already there."""
    if HEAD_FUSE == "0":
        return None

    def build(raw, residual, gamma, beta, mean, invstd):
        if not raw.is_cuda or raw.dim() != 2 or raw.shape[1] != 32 or raw.dtype != torch.float32:
            return None
        res = _native.bn_head_fwd(raw, residual, gamma, beta, mean, invstd,
                                  [W1.detach(), b1.detach(), W2.detach(), b2.detach()])
        if res is None:
            return None
        _HEAD_PREBUILT[0] = res
        return res[0]
    return build


class _Head(torch.autograd.Function):
    """sigmoid(Linear(16,1)(ELU(Linear(32,16)(emb)))) per node as one HIP kernel each way (csrc/head.hip)."""

    @staticmethod
    def forward(ctx, emb, W1, b1, W2, b2):
        pre, _HEAD_PREBUILT[0] = _HEAD_PREBUILT[0], None
        if pre is not None and pre[0].data_ptr() == emb.data_ptr() and pre[0].shape == emb.shape:
            out = pre[1]     # formed together with emb inside the BatchNorm before (head_prebuild_hook)
        else:
            out = _native.head_fwd(emb, [W1.detach(), b1.detach(), W2.detach(), b2.detach()])
        ctx.save_for_backward(emb, W1, b1, W2, b2, out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        emb, W1, b1, W2, b2, out = ctx.saved_tensors
        g_emb, gW1, gb1, gW2, gb2 = _native.head_bwd(emb, [W1.detach(), b1.detach(), W2.detach(), b2.detach()], out,
                                                     g_out.contiguous())
        return g_emb, gW1, gb1, gW2, gb2


def head(emb: torch.Tensor, W1: torch.Tensor, b1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """Per-node weight of model/net.py:46: sigmoid(W2 . ELU(W1 . emb + b1) + b2), [N]."""
    return _Head.apply(emb, W1, b1, W2, b2)
