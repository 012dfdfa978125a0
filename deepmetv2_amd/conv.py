"""EdgeConv / DynamicEdgeConv with the torch_geometric.nn signatures, backed by the HIP kernels.

Reference: `EdgeConv(nn=mesg).jittable()` constructed at /root/reference/model/graph_met_network.py:36-38 and called
at :65 (`co_conv[0](emb, edge_index)`); the dynamic-kNN alternative the north star targets is the commented line
:63.  `EdgeConv(nn=convnn, aggr=aggr)` with a multi-layer `nn` and a `.flow` read at
model/dynamic_reduction_network.py:72-73,86.

The operators own NO parameters or buffers: `nn` is the caller's module and keeps the attribute name `.nn`, so
`state_dict` keys such as `graphnet.conv_continuous.0.0.nn.0.weight` (shipped checkpoints) load unchanged.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple, Union

import torch

from . import _native
from .cluster import knn_table
from .graph import EdgeList, GraphFuture, NeighborTable, batch_info, edge_list_from_edge_index, lookup_graph
from .scatter import _SegmentMaxRows, _SegmentSumRows

_FUSED_WIDTHS = (32, 64)


def _reset(nn_module) -> None:
    """torch_geometric.nn.inits.reset restated: recurse and call reset_parameters where it exists."""
    if hasattr(nn_module, "reset_parameters"):
        nn_module.reset_parameters()
    elif hasattr(nn_module, "children"):
        for child in nn_module.children():
            _reset(child)


def _as_fusable_linear(nn_module) -> Optional[torch.nn.Linear]:
    """`nn` is exactly one Linear(2H -> H') (bare or inside a one-element Sequential), as in
    model/graph_met_network.py:36: then message+max collapses to the per-node split (csrc/edgeconv.hip)."""
    lin = None
    if isinstance(nn_module, torch.nn.Linear):
        lin = nn_module
    elif isinstance(nn_module, torch.nn.Sequential) and len(nn_module) == 1 and isinstance(nn_module[0], torch.nn.Linear):
        lin = nn_module[0]
    if lin is None or lin.in_features % 2:
        return None
    if lin.in_features // 2 not in _FUSED_WIDTHS or lin.out_features not in _FUSED_WIDTHS:
        return None
    if lin.weight.dtype != torch.float32:
        return None
    return lin


# "split": node-level dense layer (fp32 MFMA) + gather/max kernel (LDS-resident form when the events fit);
# "fused": both in one launch (edgeconv_fused_lds_kernel).  End to end the two are within 1 % of each other at
# config 2 (58 us vs 34 + 45 us per layer of a 8 ms step); "split" is the default because its gather+max kernel is
# the one BASELINE.md's roofline definition describes.
GATHER_BWD_FORM = os.environ.get("DMET_GATHER_BWD", "lds")   # "reverse": radix-sorted reverse index route
EDGECONV_FORM = os.environ.get("DMET_EDGECONV_FORM", "split")
# 160 KB LDS / 32 B per node, minus the -inf row: events up to this size gather from the LDS image.  A batch that also
# holds larger events gathers from L2 as a whole (faster than the per-event mix, see _EdgeConvLinearMax.forward).
_LDS_MAX_EVENT_NODES = int(os.environ.get("DMET_LDS_MAX_NODES", "5119"))
# the node-level dense layer of a DynamicEdgeConv rides in the kNN build's filter launch (dmet_knn_local_dense_f32)
KNN_RIDER = os.environ.get("DMET_KNN_RIDER", "1")
# the BatchNorm transform + residual add that produces a DynamicEdgeConv's input rides in that layer's kNN prep launch
BN_KNN_FUSE = os.environ.get("DMET_BN_KNN_FUSE", "1")


def _lds_eligible(x, weight, table: NeighborTable, any_size: bool = False) -> bool:
    return (x.shape[1] == 32 and weight.shape[0] == 32 and table.k in _native.LDS_GATHER_K
            and table.ptr is not None and table.max_nodes is not None
            and (any_size or table.max_nodes <= _LDS_MAX_EVENT_NODES)
            and (not table.has_int32_table() or table.nbr.data_ptr() % 16 == 0))


class _EdgeConvLinearMax(torch.autograd.Function):
    """out[i] = max_s (W.[x_i || x_j - x_i] + b), j = nbr[i,s], through P = x.(W1-W2)^T + b, Q = x.W2^T."""

    @staticmethod
    def forward(ctx, x, weight, bias, table: NeighborTable, bf16: bool = False, passthrough: bool = False):
        need_grad = any(ctx.needs_input_grad[:3])
        ctx.passthrough = passthrough
        ctx.j16 = False
        if bf16 or table.cnt is None:
            table.join()
        if bf16:
            # BASELINE configs[2]: dense layer on the bf16 matrix cores, bf16 Q table (half the gathered bytes);
            # max / add / backward stay fp32 (straight-through over the bf16 roundings)
            pq = table.pq
            table.pq = None
            if pq is not None and pq[2] == "bf16" and pq[0].shape[0] == x.shape[0]:
                P, Qh = pq[0], pq[1]   # carried by the kNN build of this x (dmet_knn_local_dense_f32, layout 2)
            else:
                P, Qh = _native.node_linear_split_bf16(x, weight, bias)
            out, arg = _native.gather_max_bf16q(P, Qh, table.nbr, want_arg=need_grad)
        elif table.cnt is not None:
            lds = (x.shape[1] == 32 and weight.shape[0] == 32 and table.ptr is not None
                   and table.max_nodes is not None and table.max_nodes <= _LDS_MAX_EVENT_NODES)
            sliced = lds and _native.GATHER_MAX_FORM != "l2-only" and os.environ.get("DMET_PQ_SLICED", "1") != "0"
            pq = table.pq
            table.pq = None
            if pq is not None and pq[2] is bool(sliced) and pq[0].shape[-2 if sliced else 0] == x.shape[0]:
                P, Q = pq[0], pq[1]     # formed together with x by the BatchNorm before (EdgeConv.prebuild_hook)
            else:
                P, Q = _native.node_linear_split(x, weight, bias, sliced=sliced)
            table.join()    # a table still being built on a side stream (graph.build_async): the dense layer ran beside it
            # radius tables with self loops (train.py:48): remember the winner's id, not its slot, so that the backward
            # needs no look-up in the 255-wide table, and walk the rows in order of their depth
            ctx.j16 = (lds and need_grad and table.nonempty and _native.GATHER_MAX_FORM == "auto"
                       and os.environ.get("DMET_RADIUS_J16", "1") != "0")
            if ctx.j16 and table.rows16 is not None and os.environ.get("DMET_RADIUS_IDS", "rows16") == "rows16":
                # ids from the event-local uint16 copy of the rows that the radius kernel wrote
                out, arg = _native.gather_max_local_j16(P, Q, table.rows16, table.cnt, table.order_by_count(), table.ptr,
                                                        table.k, sliced)
            elif ctx.j16:
                out, arg = _native.gather_max_counted_j16(P, Q, table.nbr, table.cnt, table.order_by_count(), table.ptr,
                                                          sliced)
            elif (lds and not need_grad and table.rows16 is not None and _native.GATHER_MAX_FORM == "auto"
                  and os.environ.get("DMET_RADIUS_IDS", "rows16") == "rows16"):
                # inference over a radius table: the same uint16 rows, the maximum alone
                out, arg = _native.gather_max_local_j16(P, Q, table.rows16, table.cnt, table.order_by_count(), table.ptr,
                                                        table.k, sliced, want_arg=False)
            else:
                out, arg = _native.gather_max(P, Q, table.nbr, table.ptr, want_arg=need_grad, cnt=table.cnt, lds=lds,
                                              sliced=sliced)
        elif EDGECONV_FORM == "fused" and _lds_eligible(x, weight, table):
            # gather + edge MLP + max in one launch, the event's Q slice resident in LDS
            out, arg = _native.edgeconv_fused_lds(x, weight, bias, table.nbr, table.ptr, want_arg=need_grad)
        else:
            lds = _lds_eligible(x, weight, table)
            # ragged batch with SOME events beyond the LDS image: dmet_gather_max_mixed_f32 chooses the form per event
            # inside one call.  Opt-in (DMET_GATHER_MIXED=1): measured on 64 events of 500..8000 nodes
            # (tools/gather_sweep.py) it takes 70-75 us against 35-46 us for L2 gathers on the whole batch -- the LDS
            # kernel runs one (event, slice) workgroup per CU, so ragged sizes leave CUs idle behind the largest event
            mixed = (not lds and _lds_eligible(x, weight, table, any_size=True) and table.max_nodes is not None
                     and os.environ.get("DMET_GATHER_MIXED", "0") == "1")
            # the LDS-resident gather reads P / Q slice by slice: have the dense layer write them slice-major
            sliced = lds and _native.GATHER_MAX_FORM != "l2-only" and os.environ.get("DMET_PQ_SLICED", "1") != "0"
            pq = table.pq
            table.pq = None   # one consumer: the tables must not outlive this forward inside a cached graph
            if pq is not None and pq[2] is bool(sliced) and pq[0].shape[-2 if sliced else 0] == x.shape[0]:
                P, Q = pq[0], pq[1]   # the kNN build of this x carried the dense layer (dmet_knn_local_dense_f32)
            else:
                P, Q = _native.node_linear_split(x, weight, bias, sliced=sliced)
            out, arg = _native.gather_max(P, Q, table.nbr, table.ptr, want_arg=need_grad, lds=lds,
                                          nbr_local=table.nbr_local, sliced=sliced, mixed=mixed, max_nodes=table.max_nodes)
        if need_grad:
            ctx.save_for_backward(x, weight, arg)
            ctx.table = table
            ctx.has_bias = bias is not None
        if passthrough:
            # second output: x itself, for the block's residual branch.  Its gradient then arrives HERE together with
            # g_out and is added to gx inside the backward kernel instead of by a separate autograd add
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, g_out, g_pass=None):
        x, weight, arg = ctx.saved_tensors
        table: NeighborTable = ctx.table
        H = x.shape[1]
        g_out = g_out.contiguous()
        # the node-level kernel below takes gQ slice-major: the scatter then writes contiguous runs (DMET_GQ_SLICED=0: row-major)
        fused_node = H == 32 and tuple(weight.shape) == (32, 64) and g_out.dtype == torch.float32
        gqs = fused_node and os.environ.get("DMET_GQ_SLICED", "1") != "0"
        if ctx.j16:
            # arg holds the winners' event-local ids.  A row can still be empty although the table has self loops: a
            # query with a NaN / inf coordinate finds nobody, not even itself (0xFFFF, output 0 by R3) -- the node-level
            # kernel masks g_out there from the same ids (dmet_edgeconv_linear_bwd_add_j16_f32)
            gQ = _native.gather_max_bwd_j16(g_out, arg, table.ptr, max_nodes=table.max_nodes, sliced=gqs)
        elif H == 32 and g_out.dtype == torch.float32 and table.ptr is not None and GATHER_BWD_FORM != "reverse":
            # per-event LDS scatter with exact integer sums: no reverse index (radix sort) needed
            gQ = _native.gather_max_bwd_lds(g_out, arg, table.nbr, table.ptr, nbr_local=table.nbr_local,
                                            max_nodes=table.max_nodes, sliced=gqs)
        else:
            gqs = False
            rev_ptr, rev_pos = table.reverse()
            gQ = _native.gather_max_bwd(g_out, arg, rev_ptr, rev_pos, table.k)
        if fused_node:
            # one pass over the rows: gx, gW and gb on the fp32 matrix cores (csrc/edgeconv_bwd.hip)
            gx, gW, gb = _native.edgeconv_linear_bwd(x, weight.detach(), g_out,
                                                     None if table.dense else arg, gQ,
                                                     want_bias=ctx.has_bias, g_add=g_pass, gq_sliced=gqs)
            return (gx if ctx.needs_input_grad[0] else None, gW if ctx.needs_input_grad[1] else None,
                    gb if (ctx.has_bias and ctx.needs_input_grad[2]) else None, None, None, None)
        # nodes without any neighbour produced 0 (R3): no gradient reaches P there
        none = 0xFFFF if ctx.j16 else 255
        gP = g_out if table.dense else g_out * ((arg.long() & 0xFFFF) != none).to(g_out.dtype)
        Wd = weight[:, :H] - weight[:, H:]
        W2 = weight[:, H:]
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.addmm(gP @ Wd, gQ, W2)
            if g_pass is not None:
                gx = gx + g_pass
        if ctx.needs_input_grad[1]:
            gWd = _native.xty(gP.contiguous(), x)
            gW2 = _native.xty(gQ, x)
            gW = torch.cat([gWd, gW2 - gWd], dim=1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gP.sum(0)
        return gx, gW, gb, None, None, None


def _as_mlp2(nn_module):
    """(lin1, lin2, act2, bn) when `nn` is Sequential(Linear, ELU, Linear[, ELU][, BatchNorm1d]) (ELU alpha = 1): the
    edge MLP of model/dynamic_reduction_network.py:59-70, with or without its trailing BatchNorm (bn = None); else None."""
    if not isinstance(nn_module, torch.nn.Sequential) or len(nn_module) not in (3, 4, 5):
        return None
    mods = list(nn_module)
    bn = None
    if isinstance(mods[-1], torch.nn.BatchNorm1d):
        bn = mods.pop()
        if bn.momentum is None or (not bn.track_running_stats and not bn.training):
            return None     # cumulative moving average / eval without statistics: the generic route
    if len(mods) not in (3, 4):
        return None
    l1, a1, l2 = mods[0], mods[1], mods[2]
    if not (isinstance(l1, torch.nn.Linear) and isinstance(l2, torch.nn.Linear) and isinstance(a1, torch.nn.ELU)):
        return None
    if a1.alpha != 1.0 or l1.weight.dtype != torch.float32 or l2.in_features != l1.out_features:
        return None
    act2 = False
    if len(mods) == 4:
        if not isinstance(mods[3], torch.nn.ELU) or mods[3].alpha != 1.0:
            return None
        act2 = True
    if bn is not None and bn.num_features != l2.out_features:
        return None
    return l1, l2, act2, bn


class _EdgeMLP2Bf16(torch.autograd.Function):
    """aggr_s nn([x_i || x_j - x_i]) for a two-layer nn on the bf16 matrix cores (csrc/edgemlp.hip).  The backward
    recomputes the message passing through the fp32 operators (edge features -> nn -> segment max / sum) and
    differentiates that: straight-through over the bf16 roundings, no per-edge tensor kept between the passes."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, table, act2, add, recompute, bn=None, gamma=None, beta=None):
        if bn is None:
            out = _native.edge_mlp2_bf16(x, table.nbr, W1, b1, W2, b2, act2, add)
        else:
            training = bn.training or not bn.track_running_stats
            track = bn.track_running_stats
            out = _native.edge_mlp2_bn_bf16(x, table.nbr, W1, b1, W2, b2, act2, add, gamma, beta, bn.eps, bn.momentum,
                                            bn.running_mean if track else None, bn.running_var if track else None,
                                            bn.num_batches_tracked if (track and training) else None, training)
        ctx.save_for_backward(x, W1, b1, W2, b2, gamma, beta)
        ctx.recompute = recompute
        ctx.bn = bn
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, W1, b1, W2, b2, gamma, beta = ctx.saved_tensors
        bn = ctx.bn
        # the recomputation runs the user's nn, BatchNorm module included: its running statistics were already updated by
        # the forward kernel, so the module must not move them a second time
        frozen = bn is not None and bn.training and bn.track_running_stats
        if frozen:
            momentum, tracked = bn.momentum, bn.num_batches_tracked.clone()
            bn.momentum = 0.0
        try:
            with torch.enable_grad():
                xx = x.detach().requires_grad_(True)
                out = ctx.recompute(xx)
            wanted = [(0, xx), (1, W1), (2, b1), (3, W2), (4, b2), (10, gamma), (11, beta)]
            wanted = [(i, t) for i, t in wanted if t is not None and ctx.needs_input_grad[i]]
            grads = torch.autograd.grad(out, [t for _, t in wanted], g_out.contiguous(), allow_unused=True)
        finally:
            if frozen:
                bn.momentum = momentum
                bn.num_batches_tracked.copy_(tracked)
        res = [None] * 12
        for (i, _), g in zip(wanted, grads):
            res[i] = g
        return tuple(res)


class _EdgeFeatures(torch.autograd.Function):
    """feat[e] = [x[tgt] || x[src] - x[tgt]] for a by-target grouped edge list."""

    @staticmethod
    def forward(ctx, x, edges: EdgeList):
        ctx.edges = edges
        ctx.shape = x.shape
        return _native.edge_features(x, edges.src, edges.tgt)

    @staticmethod
    def backward(ctx, g_feat):
        edges: EdgeList = ctx.edges
        srcptr, srcperm = edges.by_source()
        N, H = ctx.shape
        return _native.edge_features_bwd(g_feat.contiguous(), edges.rowptr, srcptr, srcperm, N, H), None


class EdgeConv(torch.nn.Module):
    r"""torch_geometric.nn.EdgeConv: :math:`x_i' = \mathrm{aggr}_{j \in N(i)} \; nn([x_i \,\|\, x_j - x_i])`.

    Args mirror PyG: ``nn`` (any callable mapping [*, 2F_in] -> [*, F_out]), ``aggr`` in {'max','add','sum','mean'},
    ``flow`` keyword.  ``forward(x, edge_index)`` takes a [N,F] tensor (or a pair of identical tensors) and an int64
    [2,E] edge index.
    """

    def __init__(self, nn: Callable, aggr: str = "max", **kwargs):
        super().__init__()
        flow = kwargs.pop("flow", "source_to_target")
        kwargs.pop("node_dim", None)
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")
        if aggr not in ("max", "add", "sum", "mean"):
            raise ValueError(f"unsupported aggr {aggr!r}")
        if flow not in ("source_to_target", "target_to_source"):
            raise ValueError(f"unsupported flow {flow!r}")
        self.nn = nn
        self.aggr = aggr
        self.flow = flow
        self.node_dim = 0
        # None: follow torch.autocast (bf16 autocast -> bf16 MFMA dense layer); or torch.float32 / torch.bfloat16
        self.compute_dtype = None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        _reset(self.nn)

    def jittable(self, typing: Optional[str] = None) -> "EdgeConv":
        """PyG <= 2.4 API (model/graph_met_network.py:38 calls it): nothing to specialise here."""
        return self

    def message(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        return self.nn(torch.cat([x_i, x_j - x_i], dim=-1))

    # -- the two execution paths ---------------------------------------------------------------------------
    def _forward_table(self, x: torch.Tensor, table: NeighborTable, passthrough: bool = False):
        lin = _as_fusable_linear(self.nn) if self.aggr == "max" else None
        if lin is not None and x.shape[1] * 2 == lin.in_features and table.k <= 255:  # arg slot is uint8
            if table.cnt is not None:
                self._take_prebuilt_pq(x, table)
            return _EdgeConvLinearMax.apply(x, lin.weight, lin.bias, table, self._use_bf16(lin, table), passthrough)
        table.join()
        mlp = _as_mlp2(self.nn) if self.aggr in ("max", "add", "sum") else None
        if (mlp is not None and table.cnt is None and self._wants_bf16()
                and mlp[0].in_features == 2 * x.shape[1]
                and _native.edge_mlp2_supported(x.shape[1], mlp[0].out_features, mlp[1].out_features, table.k)):
            # generic two-layer nn, bf16 compute requested: both dense layers on the matrix cores, fused with the
            # aggregation (no [E, 2H] tensor); the fp32 route below stays the default and the backward's reference
            l1, l2, act2, bn = mlp
            out = _EdgeMLP2Bf16.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, table, act2, self.aggr != "max",
                                      lambda xx: self._forward_edges(xx, table.edge_list()), bn,
                                      bn.weight if bn is not None else None, bn.bias if bn is not None else None)
            return (out, x) if passthrough else out
        out = self._forward_edges(x, table.edge_list())
        return (out, x) if passthrough else out

    def _wants_bf16(self) -> bool:
        dt = self.compute_dtype
        if dt is None and torch.is_autocast_enabled():
            dt = torch.get_autocast_gpu_dtype()
        return dt == torch.bfloat16

    def _use_bf16(self, lin: torch.nn.Linear, table: NeighborTable) -> bool:
        dt = self.compute_dtype
        if dt is None and torch.is_autocast_enabled():
            dt = torch.get_autocast_gpu_dtype()
        return (dt == torch.bfloat16 and lin.in_features == 64 and lin.out_features == 32
                and table.k in (8, 16, 32))

    def _forward_edges(self, x: torch.Tensor, edges: EdgeList) -> torch.Tensor:
        N = x.shape[0]
        if edges.num_edges == 0:
            probe = self.nn(x.new_zeros((1, 2 * x.shape[1])))
            return x.new_zeros((N, probe.shape[-1]))
        feat = _EdgeFeatures.apply(x, edges)
        msg = self.nn(feat)
        if msg.dim() != 2 or msg.shape[0] != edges.num_edges:
            raise ValueError("nn must map [E, 2F] -> [E, F_out]")
        msg = msg.contiguous()
        if self.aggr == "max":
            out, _arg = _SegmentMaxRows.apply(msg, edges.rowptr, N)
            return out
        out = _SegmentSumRows.apply(msg, edges.rowptr, N)
        if self.aggr == "mean":
            deg = (edges.rowptr[1:] - edges.rowptr[:-1]).clamp(min=1).to(out.dtype).view(-1, 1)
            out = out / deg
        return out

    def forward(self, x: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], edge_index: torch.Tensor) -> torch.Tensor:
        if isinstance(x, (tuple, list)):
            if x[1] is not None and x[1] is not x[0]:
                raise NotImplementedError("bipartite EdgeConv (x_src is not x_dst) is outside the hot path")
            x = x[0]
        if x.dim() != 2:
            raise ValueError(f"x must be [N, F], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise TypeError(f"x must be float32, got {x.dtype}")
        if isinstance(edge_index, GraphFuture):
            # built on a side stream (graph.build_async): the consumer joins as late as it can (NeighborTable.join)
            edge_index = edge_index.peek() if isinstance(edge_index.peek(), NeighborTable) else edge_index.result()
        if isinstance(edge_index, NeighborTable):
            # the table itself (dm.radius_table / dm.knn_table) instead of an edge_index tensor: same graph, but no
            # [2,E] tensor is ever sized on the host (radius_graph's exact-size result costs one sync per call)
            return self._forward_table(x, edge_index)
        hit = lookup_graph(edge_index)
        if hit is not None and hit[1] == self.flow and hit[0].num_nodes == x.shape[0]:
            return self._forward_table(x, hit[0])
        return self._forward_edges(x, edge_list_from_edge_index(edge_index, x.shape[0], self.flow))

    # -- BatchNorm transform of the PREVIOUS block fused into this layer's node-level dense layer (static graphs) --------
    def prebuild_hook(self, batch=None, graph=None):
        """A callable for dense.batch_norm(..., next_build=...) when this EdgeConv will convolve over the static `graph`
        (a NeighborTable / GraphFuture / the [2,E] tensor of radius_graph; model/graph_met_network.py:65): given the
        BatchNorm's input, residual, affine parameters and statistics it forms y = residual + BN(raw) inside the launch of
        this layer's dense layer (dmet_bn_node_linear_split_f32), keeps (P, Q) for the forward call on that y and returns
        y.  None when this layer cannot use it (DMET_BN_NLS_FUSE=0, not the fused Linear(64 -> 32) max form)."""
        if os.environ.get("DMET_BN_NLS_FUSE", "1") == "0" or self.aggr != "max" or EDGECONV_FORM != "split":
            return None
        lin = _as_fusable_linear(self.nn)
        if lin is None or lin.in_features != 64 or lin.out_features != 32 or self._wants_bf16():
            return None
        table = graph.peek() if isinstance(graph, GraphFuture) else graph
        if torch.is_tensor(table):
            hit = lookup_graph(table)
            table = hit[0] if hit is not None and hit[1] == self.flow else None
        if not isinstance(table, NeighborTable) or table.cnt is None:
            return None
        lds = table.ptr is not None and table.max_nodes is not None and table.max_nodes <= _LDS_MAX_EVENT_NODES
        sliced = bool(lds and _native.GATHER_MAX_FORM != "l2-only" and os.environ.get("DMET_PQ_SLICED", "1") != "0")

        def build(raw, residual, gamma, beta, mean, invstd):
            if not raw.is_cuda or raw.dim() != 2 or raw.shape[0] != table.num_nodes:
                return None
            out = _native.bn_node_linear_split(raw, residual, gamma, beta, mean, invstd, lin.weight.detach(),
                                               lin.bias.detach() if lin.bias is not None else None, sliced)
            if out is None:
                return None
            y, P, Q = out
            self._prebuilt_pq = (y, P, Q, sliced)
            return y
        return build

    def _take_prebuilt_pq(self, x: torch.Tensor, table: NeighborTable) -> None:
        pre, self._prebuilt_pq = getattr(self, "_prebuilt_pq", None), None
        if pre is not None and pre[0].data_ptr() == x.data_ptr() and pre[0].shape == x.shape:
            table.pq = (pre[1], pre[2], pre[3])

    def forward_with_residual_input(self, x: torch.Tensor, edge_index: torch.Tensor):
        """(conv(x), x'): x' is x routed through this operator's autograd node, for blocks of the form
        `x + f(conv(x))` (graph_met_network.py:66).  Using x' for the residual branch makes both gradients of x meet
        in this operator's backward kernel, which adds them while storing gx (no separate elementwise add)."""
        if isinstance(edge_index, GraphFuture):
            edge_index = edge_index.peek() if isinstance(edge_index.peek(), NeighborTable) else edge_index.result()
        if torch.is_tensor(x) and x.dim() == 2 and x.dtype == torch.float32:
            if isinstance(edge_index, NeighborTable):
                return self._forward_table(x, edge_index, passthrough=True)
            hit = lookup_graph(edge_index)
            if hit is not None and hit[1] == self.flow and hit[0].num_nodes == x.shape[0]:
                return self._forward_table(x, hit[0], passthrough=True)
        return self.forward(x, edge_index), x

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(nn={self.nn})"


class DynamicEdgeConv(EdgeConv):
    r"""torch_geometric.nn.DynamicEdgeConv: the graph is the k nearest neighbours of every node in the CURRENT
    feature space (self included), rebuilt on every call: ``knn(x, x, k, batch, batch).flip(0)`` upstream."""

    def __init__(self, nn: Callable, k: int, aggr: str = "max", num_workers: int = 1, **kwargs):
        super().__init__(nn=nn, aggr=aggr, **kwargs)
        if not isinstance(k, int) or k < 1:
            raise ValueError(f"k must be a positive int, got {k!r}")
        self.k = k
        self.num_workers = num_workers

    def forward(self, x: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]],
                batch: Union[None, torch.Tensor, Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
        if isinstance(x, (tuple, list)):
            if x[1] is not None and x[1] is not x[0]:
                raise NotImplementedError("bipartite DynamicEdgeConv is outside the hot path")
            x = x[0]
        if isinstance(batch, (tuple, list)):
            batch = batch[0]
        if x.dim() != 2:
            raise ValueError("Static graphs not supported in DynamicEdgeConv")  # upstream's message
        if x.dtype != torch.float32:
            raise TypeError(f"x must be float32, got {x.dtype}")
        table = self._take_prebuilt(x)
        if table is None:
            table = knn_table(x, self.k, batch, loop=True, dense=self._dense_request(x))
        return self._forward_table(x, table)

    def forward_with_residual_input(self, x: torch.Tensor, batch: Optional[torch.Tensor] = None):
        """EdgeConv.forward_with_residual_input for the dynamic graph: (conv(x), x')."""
        if x.dim() != 2 or x.dtype != torch.float32:
            return self.forward(x, batch), x
        table = self._take_prebuilt(x)
        if table is None:
            table = knn_table(x, self.k, batch, loop=True, dense=self._dense_request(x))
        return self._forward_table(x, table, passthrough=True)

    # -- BatchNorm transform of the PREVIOUS block fused into this layer's graph build --------------------------------
    def prebuild_hook(self, batch: Optional[torch.Tensor] = None, graph=None):
        """A callable for dense.batch_norm(..., next_build=...): given the BatchNorm's input, residual, affine parameters
        and batch statistics it runs this layer's graph build with the transform fused into the prep launch
        (dmet_bn_knn_local_dense_f32) and returns y = residual + BN(raw); the table is kept for the forward call on that
        y.  None when this layer cannot use it (DMET_BN_KNN_FUSE=0, k > 20, the matrix-core path switched off)."""
        if BN_KNN_FUSE == "0" or self.k > 20 or os.environ.get("DMET_KNN_PATH") == "exact":
            return None

        def build(raw, residual, gamma, beta, mean, invstd):
            if not raw.is_cuda or raw.dim() != 2 or raw.shape[1] != 32 or raw.dtype != torch.float32:
                return None
            info = batch_info(batch, raw.shape[0], raw.device)
            req = self._dense_request(raw)
            dense = (req[0], req[1], req[2](info.max_nodes)) if req is not None else None
            _native.knn_size_hint(info.min_nodes, info.max_nodes)
            out = _native.bn_knn_local_dense(raw, residual, gamma, beta, mean, invstd, info.ptr, self.k, dense)
            if out is None:
                return None
            y, nbr, dist, loc, pq = out
            table = NeighborTable(nbr, info.ptr, dense=False, dist=dist, max_nodes=info.max_nodes, nbr_local=loc)
            table.pq = pq
            self._prebuilt = (y, table)
            return y
        return build

    def _take_prebuilt(self, x: torch.Tensor):
        pre, self._prebuilt = getattr(self, "_prebuilt", None), None
        if pre is not None and pre[0].data_ptr() == x.data_ptr() and pre[0].shape == x.shape:
            return pre[1]
        return None

    def _dense_request(self, x: torch.Tensor):
        """(W, b, sliced_of(max_nodes)) when this layer runs the fused fp32 form on 32 -> 32 features: the graph build
        then carries the node-level dense layer in its filter launch (DMET_KNN_RIDER=0: its own launch, as before)."""
        if KNN_RIDER == "0" or self.aggr != "max" or EDGECONV_FORM != "split" or x.shape[1] != 32 or not x.is_cuda:
            return None
        lin = _as_fusable_linear(self.nn)
        if lin is None or lin.in_features != 64 or lin.out_features != 32 or self.k not in _native.LDS_GATHER_K:
            return None
        bf16 = self._wants_bf16()

        def sliced_of(max_nodes):
            if bf16:
                return "bf16"
            return bool(max_nodes is not None and max_nodes <= _LDS_MAX_EVENT_NODES and _native.GATHER_MAX_FORM != "l2-only"
                        and os.environ.get("DMET_PQ_SLICED", "1") != "0")
        return lin.weight, lin.bias, sliced_of

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(nn={self.nn}, k={self.k})"
