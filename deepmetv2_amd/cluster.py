"""Graph builders with the torch_cluster signatures (`knn`, `knn_graph`, `radius_graph`).

Reference call sites (relative to /root/reference): model/graph_met_network.py:63 and
model/dynamic_reduction_network.py:86,94 (knn_graph); train.py:48, evaluate.py:88, plt_weight.py:122
(radius_graph).  The kernels are in csrc/knn.hip; this file is argument checking and the int64 `edge_index` view.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _native
from .graph import NeighborTable, _deferred, batch_info

MAX_K = 64  # DMET_MAX_K


def _check_x(x: torch.Tensor) -> torch.Tensor:
    if x.dim() == 1:
        x = x.view(-1, 1)
    if x.dim() != 2:
        raise ValueError(f"x must be [N, D], got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise TypeError(f"x must be float32, got {x.dtype} (kNN distances are defined in fp32, rule R1)")
    return x


def knn_table(x: torch.Tensor, k: int, batch: Optional[torch.Tensor] = None, loop: bool = True,
              num_events: Optional[int] = None, dense=None) -> NeighborTable:
    """Fixed-width neighbour table for `x` (row i = the message sources of node i).  loop=False searches k+1
    and blanks j == i, exactly like upstream's `row != col` mask (a node whose k+1 nearest do not include itself,
    possible only with >= k+1 duplicates at lower index, keeps all k+1)."""
    x = _check_x(x)
    if not isinstance(k, int) or k < 1:
        raise ValueError(f"k must be a positive int, got {k!r}")
    kk = k if loop else k + 1
    if kk > MAX_K:
        raise ValueError(f"k={k} (searching {kk}) exceeds the supported maximum {MAX_K}")
    info = batch_info(batch, x.shape[0], x.device, num_events)
    # the LDS gather kernel reads the table as event-local uint16 ids when the kNN kernels wrote them alongside
    # dense = (W, b, sliced_of(max_nodes)): DynamicEdgeConv asks the build to carry the node-level dense layer of its
    # fused form (table.pq = (P, Q, sliced), or None when the build could not)
    loc = None
    pq = None
    _native.knn_size_hint(info.min_nodes, info.max_nodes)    # what the loader knows about the event sizes (spent by the build below)
    if loop and kk in _native.LDS_GATHER_K and dense is not None and x.shape[1] == 32:
        W, b, sliced_of = dense
        nbr, dist, loc, pq = _native.knn_local_dense(x, info.ptr, kk, W, b, sliced_of(info.max_nodes))
    elif loop and kk in _native.LDS_GATHER_K:
        nbr, dist, loc = _native.knn_local(x, info.ptr, kk)
    else:
        nbr, dist = _native.knn(x, info.ptr, kk)
    # dense <=> no -1 entry anywhere.  Sizes alone cannot promise that (a NaN query, or candidates beyond the 1e10
    # sentinel distance, leave a row short), so no table claims it.  What sizes DO give is the expectation `full_rows`:
    # self loops kept and every event >= kk nodes (batch_info / register_batch(min_nodes=...)).  The [2,E] view of such a
    # table is sized E = N kk on the host, so the reference's call shape `conv(emb, knn_graph(emb, k, batch, loop=True))`
    # (graph_met_network.py:63) enqueues without a device->host sync; the expectation is verified by a deferred check
    # (a short row would appear as -1 in the edge list, and the next operator call raises).
    dense = False
    full_rows = bool(loop and info.min_nodes is not None and info.min_nodes >= kk and x.shape[0] > 0)
    if not loop:
        self_id = torch.arange(x.shape[0], dtype=torch.int32, device=x.device).view(-1, 1)
        nbr = torch.where(nbr == self_id, torch.full_like(nbr, -1), nbr)
    table = NeighborTable(nbr, info.ptr, dense=dense, dist=dist, max_nodes=info.max_nodes, nbr_local=loc,
                          full_rows=full_rows)
    table.pq = pq
    return table


def knn_graph(x: torch.Tensor, k: int, batch: Optional[torch.Tensor] = None, loop: bool = False,
              flow: str = "source_to_target", cosine: bool = False, num_workers: int = 1,
              batch_size: Optional[int] = None) -> torch.Tensor:
    """torch_cluster.knn_graph: edge_index[2,E] int64; [0] = neighbour j, [1] = centre i for
    flow='source_to_target'; edges grouped by ascending i, ascending (distance, j) inside a group."""
    if cosine:
        raise NotImplementedError("cosine=True is not on the DeepMETv2 hot path")
    if flow not in ("source_to_target", "target_to_source"):
        raise ValueError(f"flow must be 'source_to_target' or 'target_to_source', got {flow!r}")
    _deferred.poll()
    table = knn_table(x, k, batch, loop=loop, num_events=batch_size)
    if table.full_rows and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        # rows are sorted by (d, j) with the empty slots last: the last column tells whether any row is short
        _deferred.post((table.nbr[:, -1].min() < 0).to(torch.int32),
                       "knn_graph: a neighbour row came out short although every event holds at least k nodes (non-finite "
                       "coordinates, or candidates beyond the 1e10 sentinel distance): the [2,E] edge index handed out "
                       "for it carries -1 entries")
    return table.edge_index(flow)


def knn(x: torch.Tensor, y: torch.Tensor, k: int, batch_x: Optional[torch.Tensor] = None,
        batch_y: Optional[torch.Tensor] = None, cosine: bool = False, num_workers: int = 1) -> torch.Tensor:
    """torch_cluster.knn restricted to the self-query form (y is x) that DynamicEdgeConv uses:
    returns [2,E] with row 0 = query index, row 1 = neighbour index."""
    if cosine:
        raise NotImplementedError("cosine=True is not on the DeepMETv2 hot path")
    if y is not x or (batch_y is not batch_x):
        raise NotImplementedError("knn(x, y): only the self-query form y is x is implemented (DynamicEdgeConv)")
    return knn_table(x, k, batch_x, loop=True).edge_index("target_to_source")


def radius_table(x: torch.Tensor, r: float, batch: Optional[torch.Tensor] = None, loop: bool = False,
                 max_num_neighbors: int = 32, num_events: Optional[int] = None, int32_rows: Optional[bool] = None) -> NeighborTable:
    """The radius graph as a NeighborTable.  int32_rows=False: only the event-local uint16 rows are written (what the
    fused EdgeConv reads on events of at most 65534 nodes); `.nbr` is expanded from them if somebody asks.  None: False
    when the caller registered the batch's largest event (register_batch(max_nodes=) <= 65534), else True."""
    x = _check_x(x)
    if x.shape[1] > 8:
        raise ValueError("radius_graph supports up to 8 coordinates")
    # upstream's loop=False: search max_num_neighbors + 1 and drop the node itself (done inside the kernel)
    m = max_num_neighbors if loop else max_num_neighbors + 1
    info = batch_info(batch, x.shape[0], x.device, num_events)
    # no -1 fill of the unused slots: every consumer of a table with `cnt` goes by cnt
    if int32_rows is None:
        int32_rows = not (info.max_nodes is not None and info.max_nodes <= 65534
                          and os.environ.get("DMET_RADIUS_INT32", "lazy") == "lazy")
    nbr, _cnt, rows16 = _native.radius(x, info.ptr, r, m, skip_self=not loop, pad=False, local=True, int32_rows=int32_rows)
    # with self loops every node finds at least itself (the cap counts hits in index order, but a full row is not empty)
    return NeighborTable(nbr, info.ptr, dense=False, max_nodes=info.max_nodes, cnt=_cnt, nonempty=bool(loop),
                         rows16=rows16, shape=(x.shape[0], m))


def radius_graph(x: torch.Tensor, r: float, batch: Optional[torch.Tensor] = None, loop: bool = False,
                 max_num_neighbors: int = 32, flow: str = "source_to_target", num_workers: int = 1,
                 batch_size: Optional[int] = None) -> torch.Tensor:
    """torch_cluster.radius_graph (train.py:48 passes r=0.4, loop=True, max_num_neighbors=255)."""
    if flow not in ("source_to_target", "target_to_source"):
        raise ValueError(f"flow must be 'source_to_target' or 'target_to_source', got {flow!r}")
    # the [2,E] view is cut from the int32 table: have the build write it (radius_table alone leaves it out when it can)
    return radius_table(x, r, batch, loop, max_num_neighbors, batch_size, int32_rows=True).edge_index(flow)
