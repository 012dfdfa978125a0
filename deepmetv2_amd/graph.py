"""Graph containers shared by the operators: ragged batch info, fixed-width neighbour tables, CSR edge lists.

The PyG-facing API speaks `batch[N]` / `edge_index[2,E]` int64 tensors (the reference's call sites pass exactly
those: /root/reference/train.py:48-49, model/graph_met_network.py:63-65).  Internally the kernels want `ptr[B+1]`
and either a fixed-width table nbr[N,k] (what our own kNN / radius kernels emit) or a by-target CSR.  The
registries below let an `edge_index` produced by `knn_graph` / `radius_graph` find its table again when it is
handed to `EdgeConv.forward`, so the hot path never re-derives structure from the int64 edge list.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch

from . import _native


# ---------------------------------------------------------------------------------------------------------------
# batch vector -> (ptr, B)
# ---------------------------------------------------------------------------------------------------------------
class BatchInfo:
    __slots__ = ("ptr", "num_events", "num_nodes", "max_nodes", "min_nodes")

    def __init__(self, ptr: torch.Tensor, num_events: int, num_nodes: int, max_nodes: Optional[int] = None,
                 min_nodes: Optional[int] = None):
        self.ptr = ptr
        self.num_events = num_events
        self.num_nodes = num_nodes
        self.max_nodes = max_nodes      # largest event (None = unknown): picks the LDS-resident EdgeConv kernel
        self.min_nodes = min_nodes      # smallest event (None = unknown): >= k means a kNN table without empty slots


_batch_registry: Dict[int, Tuple[weakref.ref, int, BatchInfo]] = {}


def _registry_get(reg, t: torch.Tensor):
    ent = reg.get(id(t))
    if ent is None:
        return None
    ref, version, val = ent
    if ref() is t and t._version == version:
        return val
    del reg[id(t)]
    return None


def _registry_put(reg, t: torch.Tensor, val) -> None:
    key = id(t)

    def _drop(_ref, key=key, reg=reg):
        ent = reg.get(key)
        if ent is not None and ent[0] is _ref:
            del reg[key]

    reg[key] = (weakref.ref(t, _drop), t._version, val)


def register_batch(batch: torch.Tensor, ptr: torch.Tensor, num_events: int,
                   max_nodes: Optional[int] = None, min_nodes: Optional[int] = None) -> BatchInfo:
    """Tell the operators the ptr / event count (/ largest and smallest event) of a batch vector up front (avoids a
    device->host sync on first use; without `max_nodes` one sync happens here, which then also yields `min_nodes`)."""
    ptr = ptr.to(torch.int64).contiguous()
    if max_nodes is None and ptr.numel() > 1:
        d = ptr.diff()
        max_nodes, mn = (int(v) for v in torch.stack([d.max(), d.min()]).tolist())
        min_nodes = mn if min_nodes is None else min_nodes
    info = BatchInfo(ptr, int(num_events), int(batch.numel()), max_nodes, min_nodes)
    _registry_put(_batch_registry, batch, info)
    return info


def batch_info(batch: Optional[torch.Tensor], num_nodes: int, device: torch.device,
               num_events: Optional[int] = None) -> BatchInfo:
    if batch is None:
        ptr = torch.tensor([0, num_nodes], dtype=torch.int64, device=device)
        return BatchInfo(ptr, 1, num_nodes, num_nodes, num_nodes)
    if batch.dim() != 1 or batch.numel() != num_nodes:
        raise ValueError(f"batch must be 1-D with {num_nodes} entries, got {tuple(batch.shape)}")
    info = _registry_get(_batch_registry, batch)
    if info is not None:
        return info
    if batch.dtype != torch.int64:
        raise TypeError(f"batch must be int64 (torch.long), got {batch.dtype}")
    if num_nodes == 0:
        return BatchInfo(torch.zeros(1, dtype=torch.int64, device=device), 0, 0)
    # one host sync, like PyG's own `int(batch.max()) + 1`; sortedness is a documented precondition upstream
    if num_events is None:
        last, unsorted = torch.stack([batch[-1], (batch[1:] < batch[:-1]).any().to(batch.dtype)]).tolist()
        if unsorted:
            raise ValueError("batch vector must be sorted (torch_cluster / PyG precondition)")
        num_events = int(last) + 1
    ptr = _native.batch_to_ptr(batch, num_events)
    mx = mn = 0
    if num_events > 0:
        d = ptr.diff()
        mx, mn = (int(v) for v in torch.stack([d.max(), d.min()]).tolist())
    info = BatchInfo(ptr, num_events, num_nodes, mx, mn)
    _registry_put(_batch_registry, batch, info)
    return info


# ---------------------------------------------------------------------------------------------------------------
# deferred device-side checks (the CUDA convention: an error found on the device surfaces at a later host call)
# ---------------------------------------------------------------------------------------------------------------
class _DeferredChecks:
    """Conditions that hold for every sane input but can only be verified on the device are checked WITHOUT stalling the
    launch thread: the device writes a flag, a non-blocking copy brings it to pinned host memory, and a later call of the
    operators (or `deepmetv2_amd.raise_deferred_errors()`, which waits) looks at the copies that have completed."""
    SLOTS = 64

    def __init__(self):
        self.host = None
        self.pending = []       # (event, slot, message)
        self.next = 0

    def post(self, flag: torch.Tensor, message: str) -> None:
        """flag: 0-d / 1-element int32 device tensor, non-zero = the condition was violated."""
        if self.host is None:
            self.host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        slot = self.next % self.SLOTS
        self.next += 1
        for ent in [p for p in self.pending if p[1] == slot]:     # the ring came round: this copy is 64 builds old
            ent[0].synchronize()
            self._settle(ent)
        self.host[slot:slot + 1].copy_(flag.reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, slot, message))

    def _settle(self, ent) -> None:
        self.pending.remove(ent)
        if int(self.host[ent[1]]) != 0:
            raise RuntimeError(ent[2])

    def poll(self, wait: bool = False) -> None:
        for ent in list(self.pending):
            if wait:
                ent[0].synchronize()
            if wait or ent[0].query():
                self._settle(ent)


_deferred = _DeferredChecks()


def raise_deferred_errors() -> None:
    """Wait for every outstanding device-side check and raise if one failed (call it where the training loop
    synchronises anyway, e.g. next to `loss.item()`, train.py:54)."""
    _deferred.poll(wait=True)


# ---------------------------------------------------------------------------------------------------------------
# CSR edge list (by target), plus the by-source index for the backward pass
# ---------------------------------------------------------------------------------------------------------------
class EdgeList:
    """Edges grouped by target: tgt[e] == i for rowptr[i] <= e < rowptr[i+1]; src/tgt int32."""

    def __init__(self, src: torch.Tensor, tgt: torch.Tensor, rowptr: torch.Tensor, num_nodes: int,
                 perm: Optional[torch.Tensor] = None):
        self.src = src
        self.tgt = tgt
        self.rowptr = rowptr
        self.num_nodes = num_nodes
        self.num_edges = int(src.numel())
        self.perm = perm  # position in the caller's edge_index of each grouped edge (None = identity)
        self._by_source = None

    def by_source(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """srcptr[N+1], srcperm[E]: edges leaving node j, ascending edge position (deterministic backward)."""
        if self._by_source is None:
            if self.num_edges == 0:
                z = torch.zeros(self.num_nodes + 1, dtype=torch.int32, device=self.src.device)
                self._by_source = (z, torch.zeros(1, dtype=torch.int32, device=self.src.device))
            else:
                self._by_source = _reverse_of_column(self.src, self.num_nodes)
        return self._by_source


def _reverse_of_column(col: torch.Tensor, num_nodes: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stable sort of positions 0..E-1 by col value -> (ptr[num_nodes+1], perm[E]) via the reverse-index kernel."""
    return _native.reverse_index(col.contiguous(), num_nodes)


def edge_list_from_edge_index(edge_index: torch.Tensor, num_nodes: int, flow: str) -> EdgeList:
    """Generic path for an arbitrary caller-supplied edge_index (radius graphs, to_undirected output...)."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    if edge_index.dtype != torch.int64:
        raise TypeError(f"edge_index must be int64 (torch.long), got {edge_index.dtype}")
    i_row, j_row = (1, 0) if flow == "source_to_target" else (0, 1)
    tgt64 = edge_index[i_row]
    src64 = edge_index[j_row]
    E = tgt64.numel()
    dev = edge_index.device
    if E == 0:
        z = torch.zeros(num_nodes + 1, dtype=torch.int32, device=dev)
        e = torch.zeros(0, dtype=torch.int32, device=dev)
        return EdgeList(e, e, z, num_nodes)
    tgt = tgt64.to(torch.int32).contiguous()
    src = src64.to(torch.int32).contiguous()
    # group by target with the stable reverse-index sort (keeps the caller's order inside a group, so "lowest
    # edge position wins ties" (R4) is preserved); skip the permutation when the list is already grouped.
    rowptr, perm = _reverse_of_column(tgt, num_nodes)
    ident = torch.arange(E, dtype=torch.int32, device=dev)
    if bool((perm[:E] == ident).all()):
        return EdgeList(src, tgt, rowptr, num_nodes, None)
    p = perm[:E].to(torch.int64)
    return EdgeList(src[p].contiguous(), tgt[p].contiguous(), rowptr, num_nodes, perm[:E])


# ---------------------------------------------------------------------------------------------------------------
# fixed-width neighbour table (what the kNN / radius kernels emit)
# ---------------------------------------------------------------------------------------------------------------
class NeighborTable:
    """nbr[N,k] int32 global node ids, -1 = empty slot.  Row i lists the message SOURCES of target i.
    With `cnt` (radius tables) only the first cnt[i] slots of row i are defined; the rest may be unwritten."""

    def __init__(self, nbr: Optional[torch.Tensor], ptr: Optional[torch.Tensor], dense: bool, dist: Optional[torch.Tensor] = None,
                 max_nodes: Optional[int] = None, cnt: Optional[torch.Tensor] = None,
                 nbr_local: Optional[torch.Tensor] = None, nonempty: bool = False,
                 rows16: Optional[torch.Tensor] = None, full_rows: bool = False, shape: Optional[Tuple[int, int]] = None):
        # nbr=None with `shape` = (N, k), `rows16`, `cnt` and `ptr`: a counted table that exists as event-local uint16 rows
        # only (cluster.radius_table); the int32 form is expanded from them the first time somebody asks for `.nbr`
        if nbr is None and (shape is None or rows16 is None or cnt is None or ptr is None):
            raise ValueError("NeighborTable without an int32 table needs shape, rows16, cnt and ptr")
        self._nbr = nbr
        # True: every row is EXPECTED to hold k entries (kNN with self loops, every event >= k nodes): the [2,E] view is
        # sized E = N k without asking the device; the expectation is verified by a deferred check (knn_table).  Unlike
        # `dense` it is not relied upon for masking: a short row (non-finite query) still yields 0 and no gradient.
        self.full_rows = full_rows
        self._pending = None        # a GraphFuture whose side-stream build has not been joined on the consumer's stream
        self.pq = None              # (P, Q, sliced) of the consuming EdgeConv's dense layer when the kNN build carried it
        self.rows16 = rows16        # counted tables: the rows again as event-local uint16 ids (_native.radius(local=True))
        self.nonempty = nonempty    # True: every row holds at least one entry (tables built with self loops)
        self._order = None
        self.nbr_local = nbr_local  # optional int16-typed [N,k]: the same table as event-local uint16 ids (knn_local)
        self.ptr = ptr
        self.max_nodes = max_nodes
        self.cnt = cnt              # optional int32 [N]: slots beyond cnt[i] in row i are all -1 (wide, shallow tables)
        self.num_nodes, self.k = (nbr.shape if nbr is not None else shape)
        self.dense = dense          # True: no -1 entries anywhere (every row has exactly k neighbours)
        self.dist = dist
        self._rev = None
        self._rp = None
        self._edges = None
        self._edge_index = {}

    @property
    def nbr(self) -> torch.Tensor:
        if self._nbr is None:
            # global ids of the first cnt[i] slots of row i, -1 beyond (the uint16 rows hold event-local ids; slots past
            # the last started chunk of 8 are unwritten)
            self.join()
            N, k = self.num_nodes, self.k
            counts = (self.ptr[1:] - self.ptr[:-1])
            lo = torch.repeat_interleave(self.ptr[:-1], counts, output_size=N).to(torch.int32).view(-1, 1)
            loc = self.rows16[:, :k].to(torch.int32) & 0xFFFF
            slot = torch.arange(k, device=loc.device, dtype=torch.int32).view(1, -1)
            self._nbr = torch.where(slot < self.cnt.view(-1, 1), loc + lo, torch.full_like(loc, -1)).contiguous()
        return self._nbr

    def has_int32_table(self) -> bool:
        return self._nbr is not None

    def join(self) -> "NeighborTable":
        """Make the table's device tensors usable on the current stream (a table built by graph.build_async; no-op else)."""
        if self._pending is not None:
            self._pending.result()
        return self

    def order_by_count(self) -> Optional[torch.Tensor]:
        """Counted tables: per event, the local node indices grouped by slot count (one small kernel, cached)."""
        self.join()
        if self._order is None and self.cnt is not None and self.ptr is not None:
            self._order = _native.table_order_by_count(self.cnt, self.ptr)
        return self._order

    def reverse(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """rev_ptr[N+1], rev_slot[...]: the table positions i*k+s that hold node j, ascending, for every j."""
        self.join()
        if self._rev is None:
            if self.cnt is None:
                self._rev = _native.reverse_index(self.nbr.view(-1), self.num_nodes)
            else:
                # wide, shallow table: sort the valid entries only (a 255-wide radius table is ~85 % padding)
                flat = self.nbr.view(-1)
                slot = torch.arange(self.k, device=flat.device, dtype=torch.int32).view(1, -1)
                pos = torch.nonzero((slot < self.cnt.view(-1, 1)).view(-1)).view(-1).to(torch.int32)
                keys = flat[pos.long()].contiguous()
                rev_ptr, rev_e = _native.reverse_index(keys, self.num_nodes)
                self._rev = (rev_ptr, pos[rev_e[: keys.numel()].long()].contiguous())
        return self._rev

    def _rowptr(self):
        """(rowptr[N+1] int32, E): one host sync to learn the edge count (upstream returns exact-size tensors too)."""
        self.join()
        if self._rp is None:
            if self.dense or self.full_rows:
                N, k = self.num_nodes, self.k
                self._rp = (torch.arange(0, (N + 1) * k, k, dtype=torch.int32, device=self.ptr.device if self._nbr is None else self._nbr.device), N * k)
            else:
                rp = _native.table_rowptr(self.nbr, self.cnt)
                self._rp = (rp, int(rp[-1].item()) if self.num_nodes else 0)
        return self._rp

    def edge_list(self) -> EdgeList:
        if self._edges is None:
            rowptr, E = self._rowptr()
            _ei, src, tgt = _native.table_edges(self.nbr, self.cnt, rowptr, E, False, False, True)
            self._edges = EdgeList(src, tgt, rowptr, self.num_nodes)
        return self._edges

    def edge_index(self, flow: str = "source_to_target") -> torch.Tensor:
        """int64 [2,E] in PyG orientation (R5); registered so EdgeConv can find this table again."""
        # the table remembers its edge_index tensors WEAKLY: the registry below keeps the table alive for as long as the
        # tensor lives (so that EdgeConv finds it), and a strong reference back from the table would keep both alive for
        # ever -- one table + [2,E] tensor leaked per training step of the radius_graph flow (0.6 GB at 64 x 4500)
        ref = self._edge_index.get(flow)
        ei = ref() if ref is not None else None
        if ei is None:
            rowptr, E = self._rowptr()
            ei, _s, _t = _native.table_edges(self.nbr, self.cnt, rowptr, E, flow != "source_to_target", True, False)
            self._edge_index[flow] = weakref.ref(ei)
            _registry_put(_graph_registry, ei, (self, flow))
        return ei


_graph_registry: Dict[int, Tuple[weakref.ref, int, object]] = {}


class GraphFuture:
    """A graph that is being built on a side HIP stream while the caller goes on enqueueing work that does not need it.

    The reference builds its static graph first and only then calls the model (train.py:48-49), but the model's encoder
    (embeddings, three dense layers, bn_all; model/graph_met_network.py:48-58) does not depend on the graph.  The radius
    build is latency-bound (every wavefront resident at once, vector ALU ~50 % busy), so the encoder's kernels fit beside
    it: `build_async(lambda: radius_table(...))` forks a side stream for the build, and the first operator that is handed
    the future (`EdgeConv.forward`) joins it.  Inside a captured step the fork / join become graph dependencies."""
    _side = {}

    def __init__(self, build):
        dev = torch.cuda.current_device()
        side = GraphFuture._side.get(dev)
        if side is None:
            side = GraphFuture._side[dev] = torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)                       # the build's inputs are produced on the caller's stream
        with torch.cuda.stream(side):
            self._table = build()
        self._side_stream = side
        self._joined = False
        if isinstance(self._table, NeighborTable):
            self._table._pending = self          # consumers of the table's tensors join through NeighborTable.join()

    def peek(self):
        """The graph object WITHOUT joining: its host-side fields (sizes, largest event) are final, its device tensors
        are not ready on the caller's stream until `result()` / `NeighborTable.join()` ran."""
        return self._table

    def result(self):
        """The finished graph, on the caller's current stream (a device-side wait; the host does not block)."""
        if not self._joined:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._side_stream)
            t = self._table
            for name in ("_nbr", "cnt", "rows16", "nbr_local", "dist"):     # (_nbr: `.nbr` would expand a table kept as uint16 rows)
                v = getattr(t, name, None)
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)            # allocated on the side stream, consumed on this one
            self._joined = True
            if isinstance(t, NeighborTable):
                t._pending = None
        return self._table


def build_async(build) -> GraphFuture:
    """Run `build()` (e.g. `lambda: radius_table(etaphi, r, batch, ...)`) on a side stream; hand the returned future to the
    model where it expects `edge_index`."""
    return GraphFuture(build)


def lookup_graph(edge_index: torch.Tensor):
    """(NeighborTable, flow) if this exact edge_index tensor came from knn_graph / radius_graph, else None."""
    return _registry_get(_graph_registry, edge_index)


def to_undirected(edge_index: torch.Tensor, num_nodes: Optional[int] = None) -> torch.Tensor:
    """torch_geometric.utils.to_undirected for an index-only graph (the call shape of
    /root/reference/model/dynamic_reduction_network.py:86,94): add every reverse edge, then coalesce -- edges sorted by
    (row, col), duplicates removed.  int64 [2,E] in, int64 [2,E'] out; plain torch ops (host-side graph bookkeeping,
    the result takes EdgeConv's irregular-graph path)."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    if edge_index.numel() == 0:
        return edge_index
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    row = torch.cat([edge_index[0], edge_index[1]])
    col = torch.cat([edge_index[1], edge_index[0]])
    key = torch.unique(row * n + col)            # sorted ascending = (row, col) lexicographic
    return torch.stack([key // n, key % n], 0)
