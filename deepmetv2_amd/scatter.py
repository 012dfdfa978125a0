"""Segment reductions with the torch_scatter signatures, plus the fused per-event MET reduction (K4).

Reference call sites: `scatter_add(weights*px, batch)` / `scatter_add(weights*py, batch)` at
/root/reference/model/net.py:55-56 (loss) and :132-133 (metrics).  Kernels: csrc/misc.hip (event_sum_kernel),
csrc/edgeconv.hip (segment_reduce_kernel).  Sums are deterministic: a fixed-shape tree per event, no float atomics.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _native
from .graph import batch_info, edge_list_from_edge_index, _reverse_of_column


class _SegmentSum1D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, ptr):
        ctx.save_for_backward(index)
        return _native.segment_sum_1d(src, ptr)

    @staticmethod
    def backward(ctx, g_out):
        (index,) = ctx.saved_tensors
        return g_out.index_select(0, index), None, None


class _SegmentSumRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, msg, rowptr, num_rows):
        ctx.save_for_backward(rowptr)
        ctx.E = msg.shape[0]
        return _native.segment_sum(msg, rowptr, num_rows)

    @staticmethod
    def backward(ctx, g_out):
        (rowptr,) = ctx.saved_tensors
        return _native.segment_sum_bwd(g_out, rowptr, ctx.E), None, None


class _SegmentMaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, msg, rowptr, num_rows):
        out, arg = _native.segment_max(msg, rowptr, num_rows)
        ctx.save_for_backward(arg, rowptr)
        ctx.E = msg.shape[0]
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g_out, _g_arg):
        arg, rowptr = ctx.saved_tensors
        return _native.segment_max_bwd(g_out, arg, rowptr, ctx.E), None, None


def _dim_size(index: torch.Tensor, dim_size: Optional[int]) -> int:
    if dim_size is not None:
        return int(dim_size)
    return int(index.max()) + 1 if index.numel() else 0


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.scatter_add for the shapes on the hot path:
    1-D `src` with a sorted `index` (the batch vector): one deterministic segmented sum per event;
    2-D `src` [E,H] along dim 0 (aggr='add'): grouped by index with a stable sort, then per-row sums."""
    if index.dtype != torch.int64:
        raise TypeError(f"index must be int64, got {index.dtype}")
    if src.dim() == 1:
        if index.shape != src.shape:
            raise ValueError("index must have the same shape as a 1-D src")
        info = batch_info(index, src.numel(), src.device, dim_size)
        res = _SegmentSum1D.apply(src, index, info.ptr)
        if dim_size is not None and res.numel() < dim_size:
            res = torch.cat([res, res.new_zeros(dim_size - res.numel())])
        if out is not None:
            out.add_(res)
            return out
        return res
    if src.dim() == 2 and dim in (0, -2):
        idx = index if index.dim() == 1 else index[:, 0]
        n = _dim_size(idx, dim_size)
        rowptr, perm = _reverse_of_column(idx.to(torch.int32), n)
        E = idx.numel()
        ident = torch.arange(E, dtype=torch.int32, device=idx.device)
        grouped = src if bool((perm[:E] == ident).all()) else src.index_select(0, perm[:E].to(torch.int64))
        res = _SegmentSumRows.apply(grouped, rowptr, n)
        if out is not None:
            out.add_(res)
            return out
        return res
    raise NotImplementedError(f"scatter_add: unsupported call shape src{tuple(src.shape)} dim={dim}")


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim: int = 0, out=None,
                dim_size: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_scatter.scatter_max for [E,H] rows along dim 0: (out, arg); empty rows -> 0 (R3), arg = winning row
    position in `src` (lowest on ties, R4); rows with no entry report arg = E like upstream."""
    if src.dim() != 2 or dim not in (0, -2) or out is not None:
        raise NotImplementedError("scatter_max: only [E,H] along dim 0 without `out` is implemented")
    idx = index if index.dim() == 1 else index[:, 0]
    n = _dim_size(idx, dim_size)
    E = idx.numel()
    rowptr, perm = _reverse_of_column(idx.to(torch.int32), n)
    ident = torch.arange(E, dtype=torch.int32, device=idx.device)
    is_grouped = bool((perm[:E] == ident).all())
    grouped = src if is_grouped else src.index_select(0, perm[:E].to(torch.int64))
    res, arg = _SegmentMaxRows.apply(grouped, rowptr, n)
    arg64 = arg.to(torch.int64)
    if not is_grouped:
        arg64 = torch.where(arg64 >= 0, perm[:E].to(torch.int64)[arg64.clamp(min=0)], arg64)
    arg64 = torch.where(arg64 < 0, torch.full_like(arg64, E), arg64)
    return res, arg64


# ---------------------------------------------------------------------------------------------------------------
# K4: fused per-event MET
# ---------------------------------------------------------------------------------------------------------------
class _MetReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, x, ptr):
        ctx.save_for_backward(x, ptr)
        return _native.met_reduce(w, x, ptr)

    @staticmethod
    def backward(ctx, g_met):
        x, ptr = ctx.saved_tensors
        return _native.met_reduce_bwd(g_met, x, ptr), None, None


def met_reduce(weights: torch.Tensor, x: torch.Tensor, batch: Optional[torch.Tensor] = None,
               ptr: Optional[torch.Tensor] = None, num_events: Optional[int] = None) -> torch.Tensor:
    """met[b] = (sum_i w_i * x[i,0], sum_i w_i * x[i,1]) over the nodes of event b: both scatter_add calls of
    model/net.py:55-56 in one pass over w and the px/py columns.  Differentiable w.r.t. `weights`."""
    if ptr is None:
        ptr = batch_info(batch, weights.numel(), weights.device, num_events).ptr
    return _MetReduce.apply(weights, x, ptr)


class _MetLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, met, truth):
        loss, g = _native.met_loss(met, truth)
        ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, g_loss):
        (g,) = ctx.saved_tensors
        return g * g_loss, None


class _MetLossFromWeights(torch.autograd.Function):
    """loss_fn of model/net.py:49-62 as ONE autograd node: MET reduction + loss forward (two kernels), one backward kernel
    that applies the upstream gradient of the scalar loss itself (the chain of two nodes needs a [B,2] multiply)."""

    @staticmethod
    def forward(ctx, w, x, ptr, truth):
        met = _native.met_reduce(w, x, ptr)
        loss, g = _native.met_loss(met, truth)
        ctx.save_for_backward(g, x, ptr)
        return loss.view(())

    @staticmethod
    def backward(ctx, g_loss):
        g, x, ptr = ctx.saved_tensors
        return _native.met_reduce_bwd(g, x, ptr, scale=g_loss.reshape(1).float()), None, None, None


def met_loss_from_weights(weights: torch.Tensor, x: torch.Tensor, truth: torch.Tensor, batch: Optional[torch.Tensor] = None,
                          ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """met_loss(met_reduce(weights, x), truth) with the same bits, as one autograd node."""
    if ptr is None:
        ptr = batch_info(batch, weights.numel(), weights.device, truth.shape[0]).ptr
    return _MetLossFromWeights.apply(weights, x, ptr, truth)


def met_loss(met: torch.Tensor, truth: torch.Tensor) -> torch.Tensor:
    """0.5 * mean_b((met[b,0] + truth[b,0])^2 + (met[b,1] + truth[b,1])^2) (model/net.py:58-61) in one kernel."""
    return _MetLoss.apply(met, truth)
