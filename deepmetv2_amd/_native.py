"""Tensor-level wrappers over the C ABI: validate, allocate outputs / workspace from torch's caching allocator,
enqueue on the current HIP stream.  PyTorch is used here for device memory and streams only.

Every function requires ROCm device tensors and raises otherwise -- there is no CPU path in the product.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional, Tuple

import torch

from . import _lib


def _require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                "deepmetv2_amd: operator called with a non-GPU tensor. The HIP extension is the only "
                "implementation (no CPU fallback); move inputs to a ROCm device.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("deepmetv2_amd: all tensors must live on the same device")
    return dev


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev: torch.device) -> int:
    """The raw hipStream_t of torch's current stream on `dev` (one C call: torch.cuda.current_stream() builds a Stream object
    first -- 5 us each, ~13 times per training step on the launch thread)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


_NULL_CTX = contextlib.nullcontext()


def _on(dev: torch.device):
    """Context that makes `dev` the current device for a native call -- a no-op object when it already is (the usual
    case: one process per GPU): entering torch.cuda.device() costs the launch thread several microseconds per call,
    ~70 times per training step."""
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return _NULL_CTX
    return torch.cuda.device(dev)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes: int, dev: torch.device) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=dev)


class KernelTimer:
    """Optional HIP-event bracket around named native calls, recorded on the stream the kernel is launched on
    (bench.py turns it on for the roofline figure; costs two event records per call, no synchronisation)."""

    def __init__(self):
        self.enabled = False
        self.only = None        # optional set of names: bracket just these (every bracket costs two stream commands)
        self._pending = {}

    def reset(self) -> None:
        self._pending = {}

    def record(self, name: str, dev: torch.device):
        if not self.enabled or (self.only is not None and name not in self.only):
            return None
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(dev))
        self._pending.setdefault(name, []).append((a, b))
        return b

    def calibrate(self, dev: torch.device, n: int = 50) -> float:
        """Cost of an EMPTY bracket (two event records back to back on the stream), in ms: the two records are
        stream commands of their own, so a bracket reads ~2-4 us longer than the kernel inside it."""
        pairs = []
        st = torch.cuda.current_stream(dev)
        for _ in range(n):
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record(st)
            b.record(st)
            pairs.append((a, b))
        torch.cuda.synchronize(dev)
        ts = sorted(a.elapsed_time(b) for a, b in pairs)
        self.overhead_ms = ts[len(ts) // 2]
        return self.overhead_ms

    def summary(self) -> dict:
        """{name: (launches, mean_ms)} with the empty-bracket cost removed -- call after a device synchronise."""
        out = {}
        oh = getattr(self, "overhead_ms", 0.0)
        for name, pairs in self._pending.items():
            ts = [max(a.elapsed_time(b) - oh, 0.0) for a, b in pairs]
            out[name] = (len(ts), sum(ts) / max(len(ts), 1))
        return out


timer = KernelTimer()

# which gather + max kernel the last gather_max* call launched (bench.py labels its roofline block with it)
last_gather_kernel = None

# gather+max kernel form: "auto" = LDS-resident when the caller says the events fit, else L2 gathers;
# "lds" / "l2-only" force one form (experiments, tools/gather_micro.py)
GATHER_MAX_FORM = os.environ.get("DMET_GATHER_MAX_FORM", "auto")
RADIUS_FORM = os.environ.get("DMET_RADIUS", "windowed")   # "sweep": all pairs of an event (dmet_radius_f32)


# table widths the LDS-resident fp32 gather kernels are built for (k = 4 K4; 20 = the reference's own default,
# model/graph_met_network.py:63); other widths take the L2 form
LDS_GATHER_K = (8, 16, 20, 32)


def _note_gather(name: str) -> None:
    global last_gather_kernel
    last_gather_kernel = name


# ---- K1 ------------------------------------------------------------------------------------------------------
# ---- the weight-gradient sums of a backward pass in one launch (dmet_finalize_defer_begin / dmet_finalize_flush) --------
DEFER_FINALIZE = os.environ.get("DMET_DEFER_FINALIZE", "1") != "0"
_DEFER = {"active": False, "keep": [], "dev": None}


def finalize_defer_begin() -> bool:
    """From here to finalize_flush() (process-wide: autograd runs backward on a thread of its own) edgeconv_linear_bwd, encode_bwd / encode_bn_bwd and head_bwd leave
    the small second launch that sums their weight-gradient partials to ONE launch at the flush: their parameter
    gradients hold garbage until then (the harness flushes right after loss.backward()), and their workspaces are kept
    alive here.  DMET_DEFER_FINALIZE=0: a no-op (every call sums its own partials at once, as without this call)."""
    if not DEFER_FINALIZE:
        return False
    if _DEFER["active"]:
        finalize_flush()     # a deferral somebody left open (an exception between begin and flush): its sums are formed first
    _lib.check(_lib.load().dmet_finalize_defer_begin(), "dmet_finalize_defer_begin")
    _DEFER["active"], _DEFER["keep"], _DEFER["dev"] = True, [], None
    return True


def finalize_flush() -> None:
    """Form every queued weight-gradient sum (one launch on the stream of the device the queued calls ran on) and end
    the deferral; a no-op outside one."""
    if not _DEFER["active"]:
        return
    dev = _DEFER["dev"]
    L = _lib.load()
    try:
        if dev is not None and L.dmet_finalize_pending() > 0:
            with _on(dev):
                _lib.check(L.dmet_finalize_flush(_stream(dev)), "dmet_finalize_flush")
        else:
            _lib.check(L.dmet_finalize_flush(None), "dmet_finalize_flush")
    finally:
        _DEFER["active"], _DEFER["keep"], _DEFER["dev"] = False, [], None


def _defer_keep(dev: torch.device, *workspaces: torch.Tensor) -> None:
    # a queued sum reads its partials from the call's workspace at the flush: the workspace must not go back to the allocator
    if _DEFER["active"]:
        _DEFER["dev"] = dev
        _DEFER["keep"].extend(workspaces)


def knn_size_hint(min_nodes: Optional[int], max_nodes: Optional[int]) -> None:
    """Tell the NEXT kNN build on this thread what the caller knows about its event sizes (dmet_knn_size_hint)."""
    if min_nodes and max_nodes:
        _lib.check(_lib.load().dmet_knn_size_hint(int(min_nodes), int(max_nodes)), "dmet_knn_size_hint")


def _knn(x: torch.Tensor, ptr: torch.Tensor, k: int, stats: Optional[dict], want_local: bool, dense=None):
    """dense = (W[32,64], b or None, sliced): also ask the build for the node-level dense layer of the EdgeConv that
    consumes the graph (dmet_knn_local_dense_f32); a fourth result (P, Q) or None is then returned."""
    dev = _require_device(x, ptr)
    L = _lib.load()
    x = _f32c(x.detach(), "x")
    if x.dim() != 2:
        raise ValueError(f"x must be 2-D [N, D], got {tuple(x.shape)}")
    N, D = x.shape
    B = ptr.numel() - 1
    nbr = torch.empty((N, k), dtype=torch.int32, device=dev)
    dist = torch.empty((N, k), dtype=torch.float32, device=dev)
    # uint16 payload in an int16 tensor (torch has no arithmetic on uint16; the kernels only reinterpret the bytes)
    loc = torch.empty((N, k), dtype=torch.int16, device=dev) if want_local else None
    nb = L.dmet_knn_workspace_bytes(N, B, D, k)
    ws = _ws(nb, dev)
    pq = None
    asked = dense is not None
    if dense is not None and (D != 32 or N == 0 or B == 0 or tuple(dense[0].shape) != (32, 64)):
        dense = None
    _t = timer.record('knn', dev)
    with _on(dev):
        if dense is None:
            _lib.check(L.dmet_knn_local_f32(x.data_ptr(), ptr.data_ptr(), B, N, D, k, nbr.data_ptr(), dist.data_ptr(),
                                            loc.data_ptr() if want_local else None, ws.data_ptr(), ws.numel(),
                                            _stream(dev)), "dmet_knn_local_f32")
        else:
            import ctypes
            W, b, sliced = dense            # sliced: False / True, or "bf16" (P fp32, Q bf16: node_linear_split_bf16)
            W = _f32c(W.detach(), "W")
            bp = _f32c(b.detach(), "b").data_ptr() if b is not None else None
            if sliced == "bf16":
                Pt = torch.empty((N, 32), dtype=torch.float32, device=dev)
                Qt = torch.empty((N, 32), dtype=torch.bfloat16, device=dev)
                layout = 2
            else:
                PQ = torch.empty((2, 4, N, 8) if sliced else (2, N, 32), dtype=torch.float32, device=dev)
                Pt, Qt = PQ[0], PQ[1]
                layout, sliced = (1 if sliced else 0), bool(sliced)
            done = ctypes.c_int(0)
            _lib.check(L.dmet_knn_local_dense_f32(x.data_ptr(), ptr.data_ptr(), B, N, D, k, nbr.data_ptr(),
                                                  dist.data_ptr(), loc.data_ptr() if want_local else None,
                                                  W.data_ptr(), bp, layout, Pt.data_ptr(),
                                                  Qt.data_ptr(), ctypes.cast(ctypes.pointer(done), ctypes.c_void_p),
                                                  ws.data_ptr(), ws.numel(), _stream(dev)), "dmet_knn_local_dense_f32")
            if done.value:
                pq = (Pt, Qt, sliced)
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    if stats is not None and N > 0 and B > 0:
        with _on(dev):
            import ctypes
            out = (ctypes.c_int64 * 2)()
            _lib.check(L.dmet_knn_fallback_stats(ws.data_ptr(), N, B, D, k, ctypes.cast(out, ctypes.c_void_p),
                                                 _stream(dev)), "dmet_knn_fallback_stats")
            stats["flagged_tiles"], stats["flagged_queries"] = int(out[0]), int(out[1])
            stats["tiles"] = (N + 127) // 128
    if asked:
        return nbr, dist, loc, pq
    return nbr, dist, loc


def knn(x: torch.Tensor, ptr: torch.Tensor, k: int, stats: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """nbr[N,k] int32 (global ids, -1 padded), dist[N,k] fp32.  With a `stats` dict the call synchronises and stores
    stats['flagged_tiles'] / stats['flagged_queries'] (what the matrix-core path could not certify and recomputed
    exactly; diagnostics only)."""
    nbr, dist, _ = _knn(x, ptr, k, stats, False)
    return nbr, dist


def knn_local_dense(x: torch.Tensor, ptr: torch.Tensor, k: int, W: torch.Tensor, b: Optional[torch.Tensor],
                    sliced, stats: Optional[dict] = None):
    """knn_local() for the DynamicEdgeConv call shape: (nbr, dist, loc, pq) with pq = (P, Q, sliced) -- the node-level
    dense layer of node_linear_split(x, W, b, sliced) (sliced = "bf16": of node_linear_split_bf16(x, W, b)), computed by
    trailing workgroups of the build's filter launch -- or pq = None when this build took a path that cannot carry it
    (the caller then runs the dense layer itself)."""
    out = _knn(x, ptr, k, stats, True, dense=(W, b, sliced))
    return out if len(out) == 4 else (*out, None)


def knn_local(x: torch.Tensor, ptr: torch.Tensor, k: int, stats: Optional[dict] = None
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """knn() plus the same table as event-local uint16 ids (0xFFFF = empty slot; stored in an int16 tensor), written
    by the same kernels: half the id bytes for the LDS gather kernel.  Rows of events with more than 65535 nodes are
    unspecified (those events never take the LDS path)."""
    return _knn(x, ptr, k, stats, True)


def radius(x: torch.Tensor, ptr: torch.Tensor, r: float, max_nbr: int, skip_self: bool = False,
           pad: bool = True, local: bool = False, int32_rows: bool = True):
    """(nbr[N,max_nbr] int32, cnt[N] int32).  pad=False leaves the slots >= cnt[i] unwritten instead of filling them
    with -1 (the fill is most of a 255-wide table's bytes); only for consumers that go by cnt.
    local=True: a third result, the rows again as event-local uint16 ids (int16-typed [N, roundup8(max_nbr)], slots
    cnt[i] .. roundup8(cnt[i]) - 1 = 0xFFFF, the rest unwritten) for gather_max_local_j16; None when the all-pairs
    form is selected.  int32_rows=False (with local=True, pad=False, windowed form): the int32 table is not written at all
    and comes back as None -- for callers whose consumers read the uint16 rows (graph.NeighborTable expands them on demand)."""
    dev = _require_device(x, ptr)
    L = _lib.load()
    x = _f32c(x.detach(), "x")
    N, D = x.shape
    B = ptr.numel() - 1
    skip32 = bool(local and not int32_rows and not pad and RADIUS_FORM != "sweep")
    nbr = None if skip32 else torch.empty((N, max_nbr), dtype=torch.int32, device=dev)
    cnt = torch.empty((N,), dtype=torch.int32, device=dev)
    if local:
        rows16 = None
        if RADIUS_FORM != "sweep":
            stride16 = (max_nbr + 7) // 8 * 8
            rows16 = torch.empty((N, stride16), dtype=torch.int16, device=dev)
            ws = _ws(L.dmet_radius_workspace_bytes(N), dev)
            with _on(dev):
                _lib.check(L.dmet_radius_windowed_local_f32(x.data_ptr(), ptr.data_ptr(), B, N, D, float(r), max_nbr,
                                                            1 if skip_self else 0, 1 if pad else 0,
                                                            nbr.data_ptr() if nbr is not None else None,
                                                            cnt.data_ptr(), rows16.data_ptr(), stride16, ws.data_ptr(),
                                                            ws.numel(), _stream(dev)), "dmet_radius_windowed_local_f32")
            return nbr, cnt, rows16
        nbr, cnt = radius(x, ptr, r, max_nbr, skip_self, pad)
        return nbr, cnt, None
    with _on(dev):
        if RADIUS_FORM == "sweep":      # all pairs of an event (A/B and fallback)
            fn = L.dmet_radius_f32 if pad else L.dmet_radius_counted_f32
            _lib.check(fn(x.data_ptr(), ptr.data_ptr(), B, N, D, float(r), max_nbr, 1 if skip_self else 0,
                          nbr.data_ptr(), cnt.data_ptr(), _stream(dev)), "dmet_radius_f32")
        else:                            # same table, candidates windowed by the first coordinate
            ws = _ws(L.dmet_radius_workspace_bytes(N), dev)
            _lib.check(L.dmet_radius_windowed_f32(x.data_ptr(), ptr.data_ptr(), B, N, D, float(r), max_nbr,
                                                  1 if skip_self else 0, 1 if pad else 0, nbr.data_ptr(),
                                                  cnt.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                       "dmet_radius_windowed_f32")
    return nbr, cnt


# ---- K2+K3 fused -----------------------------------------------------------------------------------------------
def node_linear_split(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor],
                      sliced: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(P, Q) = (x (W1-W2)^T + b, x W2^T) as [N, Hout] tensors; sliced=True: the same values stored slice-major,
    [Hout/8][N][8], for gather_max(..., sliced=True) only (the returned tensors then have shape [Hout/8, N, 8])."""
    dev = _require_device(x, W, b)
    L = _lib.load()
    x = _f32c(x, "x"); W = _f32c(W, "W")
    N, Hin = x.shape
    Hout = W.shape[0]
    if W.shape[1] != 2 * Hin:
        raise ValueError(f"W must be [Hout, 2*Hin] = [*, {2 * Hin}], got {tuple(W.shape)}")
    if sliced and Hout % 8 != 0:
        raise ValueError("node_linear_split: sliced tables need Hout % 8 == 0")
    PQ = torch.empty((2, Hout // 8, N, 8) if sliced else (2, N, Hout), dtype=torch.float32, device=dev)
    bp = _f32c(b, "b").data_ptr() if b is not None else None
    _t = timer.record('node_linear_split', dev)
    with _on(dev):
        fn = L.dmet_node_linear_split_sliced_f32 if sliced else L.dmet_node_linear_split_f32
        _lib.check(fn(x.data_ptr(), N, Hin, Hout, W.data_ptr(), bp, PQ[0].data_ptr(), PQ[1].data_ptr(), _stream(dev)),
                   "dmet_node_linear_split_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return PQ[0], PQ[1]


def bn_node_linear_split(raw: torch.Tensor, residual: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                         mean: torch.Tensor, invstd: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor],
                         sliced: bool = False):
    """(y, P, Q): y = residual + BatchNorm(raw) (statistics given; the bits of bn_fwd's transform) and the dense layer
    node_linear_split(y, W, b, sliced) in ONE launch (dmet_bn_node_linear_split_f32), or None when the operands do not
    qualify (32 -> 32 features, 16-byte aligned vectors)."""
    dev = _require_device(raw, gamma, beta, mean, invstd, W)
    L = _lib.load()
    if raw.dim() != 2 or raw.shape[1] != 32 or tuple(W.shape) != (32, 64) or raw.dtype != torch.float32:
        return None
    raw = _f32c(raw, "raw"); W = _f32c(W, "W")
    vecs = [_f32c(t, "bn vector") for t in (gamma, beta, mean, invstd)]
    if residual is not None:
        residual = _f32c(residual, "residual")
        if residual.shape != raw.shape:
            raise ValueError("bn_node_linear_split: residual must have the shape of raw")
    bp = _f32c(b, "b") if b is not None else None
    if any(t.data_ptr() % 16 for t in vecs) or raw.data_ptr() % 16 or (residual is not None and residual.data_ptr() % 16):
        return None
    N = raw.shape[0]
    y = torch.empty_like(raw)
    PQ = torch.empty((2, 4, N, 8) if sliced else (2, N, 32), dtype=torch.float32, device=dev)
    if N == 0:
        return y, PQ[0], PQ[1]
    _t = timer.record('node_linear_split', dev)
    with _on(dev):
        _lib.check(L.dmet_bn_node_linear_split_f32(raw.data_ptr(), residual.data_ptr() if residual is not None else None,
                                                   vecs[0].data_ptr(), vecs[1].data_ptr(), vecs[2].data_ptr(),
                                                   vecs[3].data_ptr(), y.data_ptr(), N, 32, W.data_ptr(),
                                                   bp.data_ptr() if bp is not None else None, 1 if sliced else 0,
                                                   PQ[0].data_ptr(), PQ[1].data_ptr(), _stream(dev)),
                   "dmet_bn_node_linear_split_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return y, PQ[0], PQ[1]


def gather_max(P: torch.Tensor, Q: torch.Tensor, nbr: torch.Tensor, ptr: Optional[torch.Tensor],
               want_arg: bool, cnt: Optional[torch.Tensor] = None, lds: bool = False,
               nbr_local: Optional[torch.Tensor] = None, sliced: bool = False, mixed: bool = False,
               max_nodes: Optional[int] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """out = P + max over the rows of Q listed in nbr (+ uint8 arg).  lds=True: the caller knows every event fits
    the LDS image (<= 5119 nodes, k in LDS_GATHER_K, H % 8 == 0) -> LDS-resident kernel; mixed=True: some events do not,
    the form is chosen per event inside the call (row-major P / Q); else gathers come from L2.  max_nodes: the batch's
    largest event when the caller knows it (a hint: batches of small events run two 512-thread workgroups per CU)."""
    dev = _require_device(P, Q, nbr)
    L = _lib.load()
    if mixed and cnt is None and not sliced and ptr is not None and GATHER_MAX_FORM == "auto":
        N, H = P.shape
        k = nbr.shape[1]
        out = torch.empty((N, H), dtype=torch.float32, device=dev)
        arg = torch.empty((N, H), dtype=torch.uint8, device=dev) if want_arg else None
        if nbr_local is not None and (nbr_local.shape != nbr.shape or nbr_local.dtype != torch.int16
                                      or not nbr_local.is_contiguous()):
            raise ValueError("nbr_local must be the contiguous int16 [N, k] table of knn_local()")
        _t = timer.record('gather_max', dev)
        _note_gather("gather_max_lds_kernel (events <= 5119 nodes: Q slice resident in LDS) + gather_max_mlp_kernel "
                     "(larger events: row gathers from L2), chosen per event in one call; row-major P/Q")
        with _on(dev):
            _lib.check(L.dmet_gather_max_mixed_f32(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(),
                                                   nbr_local.data_ptr() if nbr_local is not None else None,
                                                   ptr.data_ptr(), ptr.numel() - 1, N, k, H, out.data_ptr(),
                                                   arg.data_ptr() if want_arg else None, _stream(dev)),
                       "dmet_gather_max_mixed_f32")
        if _t is not None:
            _t.record(torch.cuda.current_stream(dev))
        return out, arg
    if sliced:      # [H/8, N, 8] tables of node_linear_split(..., sliced=True)
        N, H = P.shape[1], P.shape[0] * 8
    else:
        N, H = P.shape
    k = nbr.shape[1]
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    arg = torch.empty((N, H), dtype=torch.uint8, device=dev) if want_arg else None
    if sliced and cnt is None:
        if ptr is None or k not in LDS_GATHER_K or GATHER_MAX_FORM == "l2-only":
            raise ValueError("gather_max: slice-major tables are only read by the LDS-resident kernels")
        _t = timer.record('gather_max', dev)
        _note_gather("gather_max_lds_kernel (per-event Q slice resident in LDS; slice-major P/Q, "
                     + ("uint16 event-local ids)" if nbr_local is not None else "int32 ids)"))
        with _on(dev):
            _lib.check(L.dmet_gather_max_lds_sliced_cap_f32(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(),
                                                            nbr_local.data_ptr() if nbr_local is not None else None,
                                                            ptr.data_ptr(), ptr.numel() - 1, N, k, H, out.data_ptr(),
                                                            arg.data_ptr() if want_arg else None, int(max_nodes or 0),
                                                            _stream(dev)),
                       "dmet_gather_max_lds_sliced_cap_f32")
        if _t is not None:
            _t.record(torch.cuda.current_stream(dev))
        return out, arg
    if cnt is not None:
        _t = timer.record('gather_max', dev)
        with _on(dev):
            if lds and ptr is not None and H % 8 == 0 and GATHER_MAX_FORM != "l2-only":
                _note_gather("gather_max_lds_kernel, counted rows (radius table; Q slice resident in LDS"
                             + (", slice-major P/Q)" if sliced else ")"))
                _lib.check(L.dmet_gather_max_counted_lds_f32(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(), cnt.data_ptr(),
                                                             ptr.data_ptr(), ptr.numel() - 1, N, k, H,
                                                             1 if sliced else 0, out.data_ptr(),
                                                             arg.data_ptr() if want_arg else None, _stream(dev)),
                           "dmet_gather_max_counted_lds_f32")
            elif sliced:
                raise ValueError("gather_max: slice-major tables are only read by the LDS-resident kernels")
            else:
                _note_gather("gather_max_kernel, counted rows (radius table; gathers from L2)")
                _lib.check(L.dmet_gather_max_counted_f32(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(), cnt.data_ptr(), N, k,
                                                         H, out.data_ptr(), arg.data_ptr() if want_arg else None,
                                                         _stream(dev)), "dmet_gather_max_counted_f32")
        if _t is not None:
            _t.record(torch.cuda.current_stream(dev))
        return out, arg
    _t = timer.record('gather_max', dev)
    use_lds = (lds or GATHER_MAX_FORM == "lds") and GATHER_MAX_FORM != "l2-only" and ptr is not None and H % 8 == 0
    with _on(dev):
        if use_lds and nbr_local is not None and k in LDS_GATHER_K:
            if nbr_local.shape != nbr.shape or nbr_local.dtype != torch.int16 or not nbr_local.is_contiguous():
                raise ValueError("nbr_local must be the contiguous int16 [N, k] table of knn_local()")
            _note_gather("gather_max_lds_kernel (per-event Q slice resident in LDS; row-major P/Q, uint16 ids)")
            _lib.check(L.dmet_gather_max_lds16_f32(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(), nbr_local.data_ptr(),
                                                   ptr.data_ptr(), ptr.numel() - 1, N, k, H, out.data_ptr(),
                                                   arg.data_ptr() if want_arg else None, _stream(dev)),
                       "dmet_gather_max_lds16_f32")
        else:
            fn = L.dmet_gather_max_lds_f32 if use_lds else L.dmet_gather_max_f32
            _note_gather("gather_max_lds_kernel (per-event Q slice resident in LDS; row-major P/Q, int32 ids)" if use_lds
                         else "gather_max_mlp_kernel (row gathers from L2; events too large for the LDS image)")
            _lib.check(fn(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(), ptr.data_ptr() if ptr is not None else None,
                          (ptr.numel() - 1) if ptr is not None else 0, N, k, H, out.data_ptr(),
                          arg.data_ptr() if want_arg else None, _stream(dev)), "dmet_gather_max_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out, arg


def table_order_by_count(cnt: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    """order[N] int32: per event, its local node indices grouped by slot count (deepest rows first)."""
    dev = _require_device(cnt, ptr)
    L = _lib.load()
    N = cnt.numel()
    order = torch.empty((N,), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_table_order_by_count(cnt.data_ptr(), ptr.data_ptr(), ptr.numel() - 1, N, order.data_ptr(),
                                               _stream(dev)), "dmet_table_order_by_count")
    return order


def gather_max_counted_j16(P: torch.Tensor, Q: torch.Tensor, nbr: torch.Tensor, cnt: torch.Tensor,
                           order: Optional[torch.Tensor], ptr: torch.Tensor, sliced: bool):
    """(out[N,H], argj[N,H] int16-typed uint16 winner ids) of the counted LDS gather (radius tables)."""
    dev = _require_device(P, Q, nbr, cnt, ptr)
    L = _lib.load()
    if sliced:
        N, H = P.shape[1], P.shape[0] * 8
    else:
        N, H = P.shape
    k = nbr.shape[1]
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    argj = torch.empty((N, H), dtype=torch.int16, device=dev)
    _t = timer.record('gather_max', dev)
    _note_gather("gather_max_lds_kernel, counted rows (radius table; Q slice resident in LDS, winner ids, rows ordered by "
                 "depth" + (", slice-major P/Q)" if sliced else ")"))
    with _on(dev):
        _lib.check(L.dmet_gather_max_counted_lds_j16_f32(P.data_ptr(), Q.data_ptr(), nbr.data_ptr(), cnt.data_ptr(),
                                                         order.data_ptr() if order is not None else None, ptr.data_ptr(),
                                                         ptr.numel() - 1, N, k, H, 1 if sliced else 0, out.data_ptr(),
                                                         argj.data_ptr(), _stream(dev)), "dmet_gather_max_counted_lds_j16_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out, argj


def gather_max_local_j16(P: torch.Tensor, Q: torch.Tensor, rows16: torch.Tensor, cnt: torch.Tensor,
                         order: Optional[torch.Tensor], ptr: torch.Tensor, kmax: int, sliced: bool, want_arg: bool = True):
    """gather_max_counted_j16 reading the ids from the uint16 rows of radius(..., local=True): identical (out, argj);
    want_arg=False (inference): (out, None)."""
    dev = _require_device(P, Q, rows16, cnt, ptr)
    L = _lib.load()
    if sliced:
        N, H = P.shape[1], P.shape[0] * 8
    else:
        N, H = P.shape
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    argj = torch.empty((N, H), dtype=torch.int16, device=dev) if want_arg else None
    _t = timer.record('gather_max', dev)
    _note_gather("gather_max_lds_kernel, counted rows (radius table as event-local uint16 rows; Q slice resident in LDS, "
                 "winner ids, rows ordered by depth" + (", slice-major P/Q)" if sliced else ")"))
    with _on(dev):
        _lib.check(L.dmet_gather_max_local_j16_f32(P.data_ptr(), Q.data_ptr(), rows16.data_ptr(), rows16.shape[1],
                                                   cnt.data_ptr(), order.data_ptr() if order is not None else None,
                                                   ptr.data_ptr(), ptr.numel() - 1, N, kmax, H, 1 if sliced else 0,
                                                   out.data_ptr(), argj.data_ptr() if argj is not None else None, _stream(dev)),
                   "dmet_gather_max_local_j16_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out, argj


def gather_max_bwd_j16(g_out: torch.Tensor, argj: torch.Tensor, ptr: torch.Tensor,
                       max_nodes: Optional[int] = None, sliced: bool = False) -> torch.Tensor:
    """max_nodes: the batch's largest event when the caller knows it (a hint: workgroups sized for small events).
    sliced: gQ comes back slice-major, [8, N, 4] (for edgeconv_linear_bwd(..., gq_sliced=True): a scatter workgroup then
    writes one contiguous run instead of 16-byte pieces of 128-byte rows)."""
    dev = _require_device(g_out, argj, ptr)
    L = _lib.load()
    g_out = _f32c(g_out, "g_out")
    N, H = g_out.shape
    if argj.dtype != torch.int16 or argj.shape != g_out.shape or not argj.is_contiguous():
        raise TypeError("gather_max_bwd_j16: argj must be the contiguous int16 [N,H] tensor of gather_max_counted_j16")
    sliced = bool(sliced and H == 32)
    gQ = torch.empty((8, N, 4) if sliced else (N, H), dtype=torch.float32, device=dev)
    _t = timer.record('gather_max_bwd', dev)
    with _on(dev):
        entry = L.dmet_gather_max_bwd_j16_sliced_f32 if sliced else L.dmet_gather_max_bwd_j16_cap_f32
        _lib.check(entry(g_out.data_ptr(), argj.data_ptr(), ptr.data_ptr(), ptr.numel() - 1, N, H,
                         gQ.data_ptr(), int(max_nodes or 0), _stream(dev)),
                   "dmet_gather_max_bwd_j16_sliced_f32" if sliced else "dmet_gather_max_bwd_j16_cap_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return gQ


def edgeconv_fused_lds(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], nbr: torch.Tensor,
                       ptr: torch.Tensor, want_arg: bool):
    """EdgeConv(Linear(64->32), max) in one launch (LDS-resident Q slice per event); see include/dmet.h."""
    dev = _require_device(x, W, b, nbr, ptr)
    L = _lib.load()
    x = _f32c(x, "x"); W = _f32c(W, "W")
    N, Hin = x.shape
    Hout = W.shape[0]
    k = nbr.shape[1]
    out = torch.empty((N, Hout), dtype=torch.float32, device=dev)
    arg = torch.empty((N, Hout), dtype=torch.uint8, device=dev) if want_arg else None
    bp = _f32c(b, "b").data_ptr() if b is not None else None
    _t = timer.record('edgeconv_fused', dev)
    _note_gather("edgeconv_fused_lds_kernel (gather + edge-MLP + max in one launch)")
    with _on(dev):
        _lib.check(L.dmet_edgeconv_fused_lds_f32(x.data_ptr(), nbr.data_ptr(), ptr.data_ptr(), ptr.numel() - 1, N, k,
                                                 Hin, Hout, W.data_ptr(), bp, out.data_ptr(),
                                                 arg.data_ptr() if want_arg else None, _stream(dev)),
                   "dmet_edgeconv_fused_lds_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out, arg


def edge_mlp2_supported(Hin: int, H1: int, H2: int, k: int) -> bool:
    return bool(_lib.load().dmet_edge_mlp2_supported(int(Hin), int(H1), int(H2), int(k)))


def edge_mlp2_bf16(x: torch.Tensor, nbr: torch.Tensor, W1: torch.Tensor, b1: Optional[torch.Tensor], W2: torch.Tensor,
                   b2: Optional[torch.Tensor], act2: bool, add: bool) -> torch.Tensor:
    """out[N,H2] = aggr_s nn([x_i || x_j - x_i]) for nn = Linear - ELU - Linear [- ELU] on the bf16 matrix cores,
    fused with the max / add aggregation over the fixed-width table (include/dmet.h: dmet_edge_mlp2_bf16)."""
    dev = _require_device(x, nbr, W1, W2, b1, b2)
    L = _lib.load()
    x = _f32c(x, "x"); W1 = _f32c(W1, "W1"); W2 = _f32c(W2, "W2")
    N, Hin = x.shape
    H1, H2, k = W1.shape[0], W2.shape[0], nbr.shape[1]
    if W1.shape[1] != 2 * Hin or W2.shape[1] != H1:
        raise ValueError(f"edge_mlp2: W1 must be [H1, {2 * Hin}] and W2 [H2, H1], got {tuple(W1.shape)}, {tuple(W2.shape)}")
    if nbr.dtype != torch.int32 or not nbr.is_contiguous():
        raise TypeError("edge_mlp2: nbr must be a contiguous int32 [N, k] table")
    out = torch.empty((N, H2), dtype=torch.float32, device=dev)
    _t = timer.record('edge_mlp2', dev)
    with _on(dev):
        _lib.check(L.dmet_edge_mlp2_bf16(x.data_ptr(), N, Hin, nbr.data_ptr(), k, W1.data_ptr(),
                                         _f32c(b1, "b1").data_ptr() if b1 is not None else None, H1, W2.data_ptr(),
                                         _f32c(b2, "b2").data_ptr() if b2 is not None else None, H2, 1 if act2 else 0,
                                         1 if add else 0, out.data_ptr(), _stream(dev)), "dmet_edge_mlp2_bf16")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out


def edge_mlp2_bn_bf16(x: torch.Tensor, nbr: torch.Tensor, W1: torch.Tensor, b1: Optional[torch.Tensor], W2: torch.Tensor,
                      b2: Optional[torch.Tensor], act2: bool, add: bool, gamma: Optional[torch.Tensor],
                      beta: Optional[torch.Tensor], eps: float, momentum: float, running_mean: Optional[torch.Tensor],
                      running_var: Optional[torch.Tensor], num_batches_tracked: Optional[torch.Tensor],
                      training: bool) -> torch.Tensor:
    """edge_mlp2_bf16 followed by BatchNorm1d(H2) over the edge messages, then the aggregation (include/dmet.h:
    dmet_edge_mlp2_bn_bf16); the running statistics are updated in place in training mode."""
    dev = _require_device(x, nbr, W1, W2, b1, b2, gamma, beta)
    L = _lib.load()
    x = _f32c(x, "x"); W1 = _f32c(W1, "W1"); W2 = _f32c(W2, "W2")
    N, Hin = x.shape
    H1, H2, k = W1.shape[0], W2.shape[0], nbr.shape[1]
    if W1.shape[1] != 2 * Hin or W2.shape[1] != H1:
        raise ValueError(f"edge_mlp2: W1 must be [H1, {2 * Hin}] and W2 [H2, H1], got {tuple(W1.shape)}, {tuple(W2.shape)}")
    if nbr.dtype != torch.int32 or not nbr.is_contiguous():
        raise TypeError("edge_mlp2: nbr must be a contiguous int32 [N, k] table")
    out = torch.empty((N, H2), dtype=torch.float32, device=dev)
    ptr = lambda t: _f32c(t, "param").data_ptr() if t is not None else None
    _t = timer.record('edge_mlp2', dev)
    with _on(dev):
        ws = _ws(L.dmet_edge_mlp2_bn_workspace_bytes(N, H2), dev)
        _lib.check(L.dmet_edge_mlp2_bn_bf16(x.data_ptr(), N, Hin, nbr.data_ptr(), k, W1.data_ptr(), ptr(b1), H1, W2.data_ptr(),
                                            ptr(b2), H2, 1 if act2 else 0, 1 if add else 0, ptr(gamma), ptr(beta), float(eps),
                                            float(momentum), ptr(running_mean), ptr(running_var),
                                            num_batches_tracked.data_ptr() if num_batches_tracked is not None else None,
                                            1 if training else 0, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                   "dmet_edge_mlp2_bn_bf16")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out


def node_linear_split_bf16(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16-MFMA variant: P fp32 [N,H], Q as bf16 [N,H] (gathered table)."""
    dev = _require_device(x, W, b)
    L = _lib.load()
    x = _f32c(x, "x"); W = _f32c(W, "W")
    N, Hin = x.shape
    Hout = W.shape[0]
    P = torch.empty((N, Hout), dtype=torch.float32, device=dev)
    Qh = torch.empty((N, Hout), dtype=torch.bfloat16, device=dev)
    bp = _f32c(b, "b").data_ptr() if b is not None else None
    _t = timer.record('node_linear_split', dev)
    with _on(dev):
        _lib.check(L.dmet_node_linear_split_bf16(x.data_ptr(), N, Hin, Hout, W.data_ptr(), bp, P.data_ptr(),
                                                 Qh.data_ptr(), _stream(dev)), "dmet_node_linear_split_bf16")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return P, Qh


def gather_max_bf16q(P: torch.Tensor, Qh: torch.Tensor, nbr: torch.Tensor, want_arg: bool):
    dev = _require_device(P, Qh, nbr)
    L = _lib.load()
    N, H = P.shape
    k = nbr.shape[1]
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    arg = torch.empty((N, H), dtype=torch.uint8, device=dev) if want_arg else None
    _t = timer.record('gather_max', dev)
    _note_gather("gather_max_bf16q_kernel (gather + max over the bf16 Q table, gathers from L2)")
    with _on(dev):
        _lib.check(L.dmet_gather_max_bf16q(P.data_ptr(), Qh.data_ptr(), nbr.data_ptr(), N, k, H, out.data_ptr(),
                                           arg.data_ptr() if want_arg else None, _stream(dev)), "dmet_gather_max_bf16q")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return out, arg


def gather_max_bwd(g_out: torch.Tensor, arg: torch.Tensor, rev_ptr: torch.Tensor, rev_slot: torch.Tensor,
                   k: int) -> torch.Tensor:
    dev = _require_device(g_out, arg, rev_ptr, rev_slot)
    L = _lib.load()
    g_out = _f32c(g_out, "g_out")
    N, H = g_out.shape
    gQ = torch.empty((N, H), dtype=torch.float32, device=dev)
    _t = timer.record('gather_max_bwd', dev)
    with _on(dev):
        _lib.check(L.dmet_gather_max_bwd_f32(g_out.data_ptr(), arg.data_ptr(), rev_ptr.data_ptr(),
                                             rev_slot.data_ptr(), N, k, H, gQ.data_ptr(), _stream(dev)),
                   "dmet_gather_max_bwd_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return gQ


def reverse_index(keys: torch.Tensor, num_keys: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stable sort of positions by int32 key: rev_ptr[num_keys+1] int32, rev_pos[M] int32 (see include/dmet.h)."""
    dev = _require_device(keys)
    L = _lib.load()
    if keys.dtype != torch.int32 or not keys.is_contiguous():
        raise TypeError("keys must be a contiguous int32 tensor")
    M = keys.numel()
    rev_ptr = torch.empty((num_keys + 1,), dtype=torch.int32, device=dev)
    rev_pos = torch.empty((max(M, 1),), dtype=torch.int32, device=dev)
    _t = timer.record('reverse_index', dev)
    with _on(dev):
        ws = _ws(L.dmet_reverse_index_workspace_bytes(M, num_keys), dev)
        _lib.check(L.dmet_reverse_index(keys.data_ptr(), M, num_keys, rev_ptr.data_ptr(), rev_pos.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream(dev)), "dmet_reverse_index")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return rev_ptr, rev_pos


# ---- K2 / K3 un-fused ------------------------------------------------------------------------------------------
def edge_features(x: torch.Tensor, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    dev = _require_device(x, src, tgt)
    L = _lib.load()
    x = _f32c(x, "x")
    E = src.numel()
    H = x.shape[1]
    feat = torch.empty((E, 2 * H), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_edge_features_f32(x.data_ptr(), src.data_ptr(), tgt.data_ptr(), E, H, feat.data_ptr(),
                                            _stream(dev)), "dmet_edge_features_f32")
    return feat


def edge_features_bwd(g_feat: torch.Tensor, rowptr: torch.Tensor, srcptr: torch.Tensor, srcperm: torch.Tensor,
                      N: int, H: int) -> torch.Tensor:
    dev = _require_device(g_feat, rowptr, srcptr, srcperm)
    L = _lib.load()
    g_feat = _f32c(g_feat, "g_feat")
    gx = torch.empty((N, H), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_edge_features_bwd_f32(g_feat.data_ptr(), rowptr.data_ptr(), srcptr.data_ptr(),
                                                srcperm.data_ptr(), N, H, gx.data_ptr(), _stream(dev)),
                   "dmet_edge_features_bwd_f32")
    return gx


def segment_max(msg: torch.Tensor, rowptr: torch.Tensor, N: int) -> Tuple[torch.Tensor, torch.Tensor]:
    dev = _require_device(msg, rowptr)
    L = _lib.load()
    msg = _f32c(msg, "msg")
    H = msg.shape[1]
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    arg = torch.empty((N, H), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_segment_max_f32(msg.data_ptr(), rowptr.data_ptr(), N, H, out.data_ptr(), arg.data_ptr(),
                                          _stream(dev)), "dmet_segment_max_f32")
    return out, arg


def segment_sum(msg: torch.Tensor, rowptr: torch.Tensor, N: int) -> torch.Tensor:
    dev = _require_device(msg, rowptr)
    L = _lib.load()
    msg = _f32c(msg, "msg")
    H = msg.shape[1]
    out = torch.empty((N, H), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_segment_sum_f32(msg.data_ptr(), rowptr.data_ptr(), N, H, out.data_ptr(), _stream(dev)),
                   "dmet_segment_sum_f32")
    return out


def segment_max_bwd(g_out: torch.Tensor, arg: torch.Tensor, rowptr: torch.Tensor, E: int) -> torch.Tensor:
    dev = _require_device(g_out, arg, rowptr)
    L = _lib.load()
    g_out = _f32c(g_out, "g_out")
    N, H = g_out.shape
    g_msg = torch.empty((E, H), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_segment_max_bwd_f32(g_out.data_ptr(), arg.data_ptr(), rowptr.data_ptr(), N, H,
                                              g_msg.data_ptr(), _stream(dev)), "dmet_segment_max_bwd_f32")
    return g_msg


def segment_sum_bwd(g_out: torch.Tensor, rowptr: torch.Tensor, E: int) -> torch.Tensor:
    dev = _require_device(g_out, rowptr)
    L = _lib.load()
    g_out = _f32c(g_out, "g_out")
    N, H = g_out.shape
    g_msg = torch.empty((E, H), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_segment_sum_bwd_f32(g_out.data_ptr(), rowptr.data_ptr(), N, H, g_msg.data_ptr(),
                                              _stream(dev)), "dmet_segment_sum_bwd_f32")
    return g_msg


# ---- K4 --------------------------------------------------------------------------------------------------------
def met_reduce(w: torch.Tensor, x: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    dev = _require_device(w, x, ptr)
    L = _lib.load()
    w = _f32c(w, "w")
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise TypeError("x must be float32 [N, F>=2] with unit inner stride")
    B = ptr.numel() - 1
    met = torch.empty((B, 2), dtype=torch.float32, device=dev)
    _t = timer.record('met_reduce', dev)
    with _on(dev):
        _lib.check(L.dmet_met_reduce_f32(w.data_ptr(), x.data_ptr(), x.stride(0), ptr.data_ptr(), B, met.data_ptr(),
                                         _stream(dev)), "dmet_met_reduce_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return met


def met_reduce_bwd(g_met: torch.Tensor, x: torch.Tensor, ptr: torch.Tensor,
                   scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g_w[i] = g_met[b,0] * px_i + g_met[b,1] * py_i; scale (optional, one float32 on the device) multiplies g_met first."""
    dev = _require_device(g_met, x, ptr)
    L = _lib.load()
    g_met = _f32c(g_met, "g_met")
    N = x.shape[0]
    B = ptr.numel() - 1
    g_w = torch.empty((N,), dtype=torch.float32, device=dev)
    sp = None
    if scale is not None:
        if scale.dtype != torch.float32 or scale.numel() != 1 or scale.device != dev:
            raise ValueError("met_reduce_bwd: scale must be one float32 on the device")
        sp = scale.data_ptr()
    with _on(dev):
        _lib.check(L.dmet_met_reduce_bwd_scaled_f32(g_met.data_ptr(), sp, x.data_ptr(), x.stride(0), ptr.data_ptr(), B, N,
                                                    g_w.data_ptr(), _stream(dev)), "dmet_met_reduce_bwd_scaled_f32")
    return g_w


def segment_sum_1d(src: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    dev = _require_device(src, ptr)
    L = _lib.load()
    src = _f32c(src, "src")
    B = ptr.numel() - 1
    out = torch.empty((B,), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_segment_sum_1d_f32(src.data_ptr(), ptr.data_ptr(), B, out.data_ptr(), _stream(dev)),
                   "dmet_segment_sum_1d_f32")
    return out


def batch_to_ptr(batch: torch.Tensor, B: int) -> torch.Tensor:
    dev = _require_device(batch)
    L = _lib.load()
    if batch.dtype != torch.int64:
        batch = batch.to(torch.int64)
    batch = batch.contiguous()
    ptr = torch.empty((B + 1,), dtype=torch.int64, device=dev)
    with _on(dev):
        _lib.check(L.dmet_batch_to_ptr(batch.data_ptr(), batch.numel(), B, ptr.data_ptr(), _stream(dev)),
                   "dmet_batch_to_ptr")
    return ptr


# ---- dense-layer weight gradients (N3, first piece) ---------------------------------------------------------------
def xty(A: torch.Tensor, Bm: torch.Tensor) -> torch.Tensor:
    """C[Ha,Hb] = A^T @ B for A[N,Ha], B[N,Hb] (Ha,Hb <= 64), deterministic."""
    dev = _require_device(A, Bm)
    L = _lib.load()
    A = _f32c(A, "A"); Bm = _f32c(Bm, "B")
    N, Ha = A.shape
    Hb = Bm.shape[1]
    C = torch.empty((Ha, Hb), dtype=torch.float32, device=dev)
    _t = timer.record("xty", dev)
    with _on(dev):
        ws = _ws(L.dmet_xty_workspace_bytes(N, Ha, Hb), dev)
        _lib.check(L.dmet_xty_f32(A.data_ptr(), Bm.data_ptr(), N, Ha, Hb, C.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream(dev)), "dmet_xty_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return C


def onehot_xty(index: torch.Tensor, Bm: torch.Tensor, num_rows: int) -> torch.Tensor:
    """C[R,Hb] = onehot(index)^T @ B: the weight gradient of an Embedding with R rows."""
    dev = _require_device(index, Bm)
    L = _lib.load()
    if index.dtype != torch.int64:
        raise TypeError("index must be int64")
    index = index.contiguous(); Bm = _f32c(Bm, "B")
    N, Hb = Bm.shape
    C = torch.empty((num_rows, Hb), dtype=torch.float32, device=dev)
    with _on(dev):
        ws = _ws(L.dmet_xty_workspace_bytes(N, num_rows, Hb), dev)
        _lib.check(L.dmet_onehot_xty_f32(index.data_ptr(), Bm.data_ptr(), N, num_rows, Hb, C.data_ptr(), ws.data_ptr(),
                                         ws.numel(), _stream(dev)), "dmet_onehot_xty_f32")
    return C


def _encode_params(params, dev):
    shapes = [(16, 8), (16,), (16, 24), (16,), (32, 32), (32,), (3, 8), (7, 8), (8, 8)]
    if len(params) != 9:
        raise ValueError("encode: expected 9 parameter tensors (Wc, bc, Wk, bk, Wa, ba, Echg, Epdg, Epv)")
    out = []
    for t, shp in zip(params, shapes):
        if tuple(t.shape) != shp:
            raise ValueError(f"encode: parameter of shape {tuple(t.shape)}, expected {shp}")
        if t.device != dev:
            raise ValueError("encode: parameters must be on the device of x")
        out.append(_f32c(t, "param"))
    return out


def _encode_x(x: torch.Tensor, x_cat: torch.Tensor):
    if x.dim() != 2 or x.shape[1] != 8 or x.dtype != torch.float32:
        raise ValueError(f"encode: x_cont must be float32 [N,8], got {tuple(x.shape)} {x.dtype}")
    if x_cat.dim() != 2 or x_cat.shape != (x.shape[0], 3):
        raise ValueError(f"encode: x_cat must be [N,3], got {tuple(x_cat.shape)}")
    if x_cat.dtype == torch.float32:
        # the float columns 8..10 of the same feature matrix (split_features(x, lazy_cat=True)): converted in the kernel
        if (x.stride(1) == 1 and x_cat.stride(1) == 1 and x_cat.stride(0) == x.stride(0) and x.stride(0) >= 11
                and x_cat.data_ptr() == x.data_ptr() + 32):
            return x, None
        x_cat = x_cat.long()
    if x_cat.dtype != torch.int64:
        raise ValueError(f"encode: x_cat must be int64 (or the float columns 8..10 of x), got {x_cat.dtype}")
    return (x if x.stride(1) == 1 else x.contiguous()), x_cat.contiguous()


def encode_fwd(x_cont: torch.Tensor, x_cat: torch.Tensor, params) -> torch.Tensor:
    """h[N,32] = the per-node encoder (graph_met_network.py:48-58 before bn_all) in one kernel."""
    dev = _require_device(x_cont, x_cat)
    L = _lib.load()
    x, xc = _encode_x(x_cont, x_cat)
    ps = _encode_params(params, dev)
    N = x.shape[0]
    h = torch.empty((N, 32), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_encode_fwd_f32(x.data_ptr(), x.stride(0), xc.data_ptr() if xc is not None else None, N, *[t.data_ptr() for t in ps],
                                         h.data_ptr(), _stream(dev)), "dmet_encode_fwd_f32")
    return h


def encode_bwd(x_cont: torch.Tensor, x_cat: torch.Tensor, params, h: torch.Tensor, g_h: torch.Tensor):
    """The nine parameter gradients of `encode_fwd` given its output h and dL/dh."""
    dev = _require_device(x_cont, x_cat, h, g_h)
    L = _lib.load()
    x, xc = _encode_x(x_cont, x_cat)
    ps = _encode_params(params, dev)
    h = _f32c(h, "h"); g_h = _f32c(g_h, "g_h")
    N = x.shape[0]
    grads = [torch.empty_like(t) for t in ps]
    if N == 0:
        return [g.zero_() for g in grads]
    with _on(dev):
        ws = _ws(L.dmet_encode_bwd_workspace_bytes(N), dev)
        _lib.check(L.dmet_encode_bwd_f32(x.data_ptr(), x.stride(0), xc.data_ptr() if xc is not None else None, N, *[t.data_ptr() for t in ps],
                                         h.data_ptr(), g_h.data_ptr(), *[g.data_ptr() for g in grads], ws.data_ptr(),
                                         ws.numel(), _stream(dev)), "dmet_encode_bwd_f32")
        _defer_keep(dev, ws)
    return grads


def encode_bn_bwd(x_cont: torch.Tensor, x_cat: torch.Tensor, params, h: torch.Tensor, g_y: torch.Tensor,
                  gamma: torch.Tensor, mean: torch.Tensor, invstd: torch.Tensor):
    """Backward of bn_all(encode(...)) given dL/d(bn output): (encoder grads [9], g_gamma, g_beta), the BatchNorm's
    backward transform applied inside the encoder's backward kernel (dmet_bn_bwd_stats_f32 + dmet_encode_bn_bwd_f32);
    None when nothing was launched beyond the statistics (the caller keeps the separate steps)."""
    import ctypes
    dev = _require_device(x_cont, x_cat, h, g_y)
    L = _lib.load()
    x, xc = _encode_x(x_cont, x_cat)
    ps = _encode_params(params, dev)
    h = _f32c(h, "h"); g_y = _f32c(g_y, "g_y"); gamma = _f32c(gamma.detach(), "gamma")
    N, H = h.shape
    if N == 0 or H != 32:
        return None
    grads = [torch.empty_like(t) for t in ps]
    st = torch.empty((4, H), dtype=torch.float32, device=dev)     # g_gamma, g_beta, mean_g, mean_gx
    fused = ctypes.c_int(0)
    with _on(dev):
        ws = _ws(L.dmet_bn_workspace_bytes(N, H), dev)
        _lib.check(L.dmet_bn_bwd_stats_f32(h.data_ptr(), g_y.data_ptr(), N, H, mean.data_ptr(), invstd.data_ptr(),
                                           st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                           ws.data_ptr(), ws.numel(), _stream(dev)), "dmet_bn_bwd_stats_f32")
        ws2 = _ws(L.dmet_encode_bwd_workspace_bytes(N), dev)
        _lib.check(L.dmet_encode_bn_bwd_f32(x.data_ptr(), x.stride(0), xc.data_ptr() if xc is not None else None, N,
                                            *[t.data_ptr() for t in ps], h.data_ptr(), g_y.data_ptr(), gamma.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                            *[g.data_ptr() for g in grads],
                                            ctypes.cast(ctypes.pointer(fused), ctypes.c_void_p), ws2.data_ptr(), ws2.numel(),
                                            _stream(dev)), "dmet_encode_bn_bwd_f32")
        _defer_keep(dev, ws2)
    if not fused.value:
        return None
    return grads, st[0], st[1]


def bn_fwd(x: torch.Tensor, residual: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float,
           momentum: float, running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], training: bool,
           num_batches_tracked: Optional[torch.Tensor] = None):
    """BatchNorm1d over rows (+ residual): returns (y, save_mean, save_invstd); running stats updated in place, and
    num_batches_tracked (int64 scalar on the device, training mode only) incremented by the statistics kernel."""
    dev = _require_device(x, gamma, beta)
    L = _lib.load()
    x = _f32c(x, "x")
    N, H = x.shape
    if residual is not None:
        residual = _f32c(residual, "residual")
        if residual.shape != x.shape:
            raise ValueError("bn_fwd: residual must have the shape of x")
    for t in (running_mean, running_var):
        if t is not None and (not t.is_contiguous() or t.dtype != torch.float32 or t.numel() != H):
            raise ValueError("bn_fwd: running statistics must be contiguous float32 [H]")
    gamma = _f32c(gamma, "gamma"); beta = _f32c(beta, "beta")
    y = torch.empty_like(x)
    stats = torch.empty((2, H), dtype=torch.float32, device=dev)
    with _on(dev):
        ws = _ws(L.dmet_bn_workspace_bytes(N, H), dev)
        nbt = None
        if num_batches_tracked is not None and training:
            if num_batches_tracked.dtype != torch.int64 or num_batches_tracked.numel() != 1 or num_batches_tracked.device != dev:
                raise ValueError("bn_fwd: num_batches_tracked must be an int64 scalar on x's device")
            nbt = num_batches_tracked.data_ptr()
        _lib.check(L.dmet_bn_fwd_tracked_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None, N, H,
                                             gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum),
                                             running_mean.data_ptr() if running_mean is not None else None,
                                             running_var.data_ptr() if running_var is not None else None, nbt,
                                             1 if training else 0, y.data_ptr(), stats[0].data_ptr(),
                                             stats[1].data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                   "dmet_bn_fwd_tracked_f32")
    return y, stats[0], stats[1]


def bn_stats(x: torch.Tensor, eps: float, momentum: float, running_mean: Optional[torch.Tensor],
             running_var: Optional[torch.Tensor], num_batches_tracked: Optional[torch.Tensor] = None):
    """(save_mean, save_invstd) of the training-mode BatchNorm1d over the rows of x; running statistics and
    num_batches_tracked updated like bn_fwd(training=True).  The transform is applied elsewhere (bn_knn_local_dense)."""
    dev = _require_device(x)
    L = _lib.load()
    x = _f32c(x, "x")
    N, H = x.shape
    stats = torch.empty((2, H), dtype=torch.float32, device=dev)
    with _on(dev):
        ws = _ws(L.dmet_bn_workspace_bytes(N, H), dev)
        _lib.check(L.dmet_bn_stats_f32(x.data_ptr(), N, H, float(eps), float(momentum),
                                       running_mean.data_ptr() if running_mean is not None else None,
                                       running_var.data_ptr() if running_var is not None else None,
                                       num_batches_tracked.data_ptr() if num_batches_tracked is not None else None,
                                       stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                   "dmet_bn_stats_f32")
    return stats[0], stats[1]


def _aligned16(t: torch.Tensor) -> torch.Tensor:
    """`t` itself when its storage starts on a 16-byte boundary, else a fresh (aligned) copy."""
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def bn_apply(x: torch.Tensor, residual: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
             mean: torch.Tensor, invstd: torch.Tensor) -> torch.Tensor:
    """y = (x - mean) * (gamma * invstd) + beta (+ residual): the transform of bn_fwd with the statistics given
    (bn_stats / bn_eval_stats) -- same kernel, same bits (dmet_bn_apply_f32).  Small vectors that do not start on a
    16-byte boundary (views into somebody else's flat buffer) are copied first."""
    dev = _require_device(x, gamma, beta, mean, invstd)
    L = _lib.load()
    x = _aligned16(_f32c(x, "x"))
    N, H = x.shape
    if residual is not None:
        residual = _aligned16(_f32c(residual, "residual"))
        if residual.shape != x.shape:
            raise ValueError("bn_apply: residual must have the shape of x")
    vec = [_aligned16(_f32c(t, n)) for t, n in ((gamma, "gamma"), (beta, "beta"), (mean, "mean"), (invstd, "invstd"))]
    if any(v.numel() != H for v in vec):
        raise ValueError("bn_apply: gamma / beta / mean / invstd must have H elements")
    y = torch.empty_like(x)
    with _on(dev):
        _lib.check(L.dmet_bn_apply_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None, N, H,
                                       vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(),
                                       y.data_ptr(), _stream(dev)), "dmet_bn_apply_f32")
    return y


def bn_eval_stats(running_mean: torch.Tensor, running_var: torch.Tensor, eps: float):
    """(mean, invstd) of an eval-mode BatchNorm1d: running_mean and 1 / sqrt(running_var + eps), as bn_fwd(training=False)
    forms them."""
    dev = _require_device(running_mean, running_var)
    L = _lib.load()
    H = running_mean.numel()
    stats = torch.empty((2, H), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_bn_eval_stats_f32(running_mean.data_ptr(), running_var.data_ptr(), H, float(eps),
                                            stats[0].data_ptr(), stats[1].data_ptr(), _stream(dev)), "dmet_bn_eval_stats_f32")
    return stats[0], stats[1]


def bn_knn_local_dense(raw: torch.Tensor, residual: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                       mean: torch.Tensor, invstd: torch.Tensor, ptr: torch.Tensor, k: int, dense=None):
    """y = residual + BatchNorm(raw) (statistics given) fused into the prep launch of the kNN build on y
    (dmet_bn_knn_local_dense_f32).  Returns (y, nbr, dist, loc, pq) -- pq as in knn_local_dense, None without `dense` or
    when the build could not carry the dense layer -- or None when nothing was launched (the build would not take the
    matrix-core path): the caller then applies the transform and builds the graph itself."""
    import ctypes
    dev = _require_device(raw, ptr)
    L = _lib.load()
    raw = _f32c(raw.detach(), "raw")
    N, D = raw.shape
    B = ptr.numel() - 1
    if D != 32 or N == 0 or B == 0 or k > 20:
        return None
    if residual is not None:
        residual = _f32c(residual.detach(), "residual")
    gamma = _f32c(gamma.detach(), "gamma"); beta = _f32c(beta.detach(), "beta")
    y = torch.empty_like(raw)
    nbr = torch.empty((N, k), dtype=torch.int32, device=dev)
    dist = torch.empty((N, k), dtype=torch.float32, device=dev)
    loc = torch.empty((N, k), dtype=torch.int16, device=dev)
    Wp = bp = Pp = Qp = None
    layout, Pt, Qt, sliced = 0, None, None, False
    if dense is not None and tuple(dense[0].shape) == (32, 64):
        W, b, sliced = dense
        W = _f32c(W.detach(), "W")
        Wp = W.data_ptr()
        bp = _f32c(b.detach(), "b").data_ptr() if b is not None else None
        if sliced == "bf16":
            Pt = torch.empty((N, 32), dtype=torch.float32, device=dev)
            Qt = torch.empty((N, 32), dtype=torch.bfloat16, device=dev)
            layout = 2
        else:
            PQ = torch.empty((2, 4, N, 8) if sliced else (2, N, 32), dtype=torch.float32, device=dev)
            Pt, Qt = PQ[0], PQ[1]
            layout, sliced = (1 if sliced else 0), bool(sliced)
        Pp, Qp = Pt.data_ptr(), Qt.data_ptr()
    nb = L.dmet_knn_workspace_bytes(N, B, D, k)
    ws = _ws(nb, dev)
    done, fused = ctypes.c_int(0), ctypes.c_int(0)
    _t = timer.record('knn', dev)
    with _on(dev):
        _lib.check(L.dmet_bn_knn_local_dense_f32(raw.data_ptr(), residual.data_ptr() if residual is not None else None,
                                                 gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                 y.data_ptr(), ptr.data_ptr(), B, N, D, k, nbr.data_ptr(), dist.data_ptr(),
                                                 loc.data_ptr(), Wp, bp, layout, Pp, Qp,
                                                 ctypes.cast(ctypes.pointer(done), ctypes.c_void_p),
                                                 ctypes.cast(ctypes.pointer(fused), ctypes.c_void_p), ws.data_ptr(),
                                                 ws.numel(), _stream(dev)), "dmet_bn_knn_local_dense_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    if not fused.value:
        return None
    return y, nbr, dist, loc, ((Pt, Qt, sliced) if done.value else None)


def bn_bwd(x: torch.Tensor, g_y: torch.Tensor, gamma: torch.Tensor, save_mean: torch.Tensor, save_invstd: torch.Tensor):
    """(g_x, g_gamma, g_beta) of the training-mode BatchNorm1d."""
    dev = _require_device(x, g_y, gamma)
    L = _lib.load()
    x = _f32c(x, "x"); g_y = _f32c(g_y, "g_y"); gamma = _f32c(gamma, "gamma")
    N, H = x.shape
    g_x = torch.empty_like(x)
    gg = torch.empty((2, H), dtype=torch.float32, device=dev)
    with _on(dev):
        ws = _ws(L.dmet_bn_workspace_bytes(N, H), dev)
        _lib.check(L.dmet_bn_bwd_f32(x.data_ptr(), g_y.data_ptr(), N, H, gamma.data_ptr(), save_mean.data_ptr(),
                                     save_invstd.data_ptr(), g_x.data_ptr(), gg[0].data_ptr(), gg[1].data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream(dev)), "dmet_bn_bwd_f32")
    return g_x, gg[0], gg[1]


def edgeconv_linear_bwd(x: torch.Tensor, weight: torch.Tensor, g_out: torch.Tensor, arg: Optional[torch.Tensor],
                        gQ: torch.Tensor, want_bias: bool = True, g_add: Optional[torch.Tensor] = None,
                        gq_sliced: bool = False):
    """(gx[N,32], gW[32,64], gb[32] or None) of the fused EdgeConv dense layer (H = 32) from g_out, arg and gQ;
    g_add[N,32] (optional) is added to gx inside the kernel (the residual branch's gradient).  arg: uint8 winning slots
    (255 = none) or uint16 winner ids (0xFFFF = none): g_out is masked to 0 there (R3: such a node produced 0)."""
    dev = _require_device(x, weight, g_out, gQ)
    L = _lib.load()
    x = _f32c(x, "x"); weight = _f32c(weight, "weight"); g_out = _f32c(g_out, "g_out"); gQ = _f32c(gQ, "gQ")
    N, H = x.shape
    if H != 32 or tuple(weight.shape) != (32, 64) or g_out.shape != x.shape or (
            tuple(gQ.shape) != ((8, N, 4) if gq_sliced else (N, H))):
        raise ValueError("edgeconv_linear_bwd: built for x[N,32], weight[32,64], gQ[N,32] (or slice-major [8,N,4])")
    if arg is not None and (arg.dtype not in (torch.uint8, torch.uint16, torch.int16) or arg.shape != x.shape
                            or not arg.is_contiguous()):
        raise ValueError("edgeconv_linear_bwd: arg must be contiguous uint8 (slots) or uint16 (winner ids) [N,32]")
    j16 = arg is not None and arg.dtype != torch.uint8
    gx = torch.empty_like(x)
    gW = torch.empty_like(weight)
    gb = torch.empty((H,), dtype=torch.float32, device=dev) if want_bias else None
    with _on(dev):
        ws = _ws(L.dmet_edgeconv_linear_bwd_workspace_bytes(N, H), dev)
        if g_add is not None:
            g_add = _f32c(g_add, "g_add")
            if g_add.shape != x.shape:
                raise ValueError("edgeconv_linear_bwd: g_add must have the shape of x")
        if gq_sliced:
            _lib.check(L.dmet_edgeconv_linear_bwd_sliced_f32(x.data_ptr(), weight.data_ptr(), g_out.data_ptr(),
                                                             arg.data_ptr() if arg is not None else None, int(j16), gQ.data_ptr(),
                                                             g_add.data_ptr() if g_add is not None else None, N, H,
                                                             gx.data_ptr(), gW.data_ptr(), gb.data_ptr() if gb is not None else None,
                                                             ws.data_ptr(), ws.numel(), _stream(dev)),
                       "dmet_edgeconv_linear_bwd_sliced_f32")
        else:
            entry = L.dmet_edgeconv_linear_bwd_add_j16_f32 if j16 else L.dmet_edgeconv_linear_bwd_add_f32
            _lib.check(entry(x.data_ptr(), weight.data_ptr(), g_out.data_ptr(),
                             arg.data_ptr() if arg is not None else None, gQ.data_ptr(),
                             g_add.data_ptr() if g_add is not None else None, N, H,
                             gx.data_ptr(), gW.data_ptr(),
                             gb.data_ptr() if gb is not None else None,
                             ws.data_ptr(), ws.numel(), _stream(dev)),
                       "dmet_edgeconv_linear_bwd_add_j16_f32" if j16 else "dmet_edgeconv_linear_bwd_add_f32")
        _defer_keep(dev, ws)
    return gx, gW, gb


def gather_max_bwd_lds(g_out: torch.Tensor, arg: torch.Tensor, nbr: torch.Tensor, ptr: torch.Tensor,
                       nbr_local: Optional[torch.Tensor] = None, max_nodes: Optional[int] = None,
                       sliced: bool = False) -> torch.Tensor:
    """gQ[N,32] by per-event LDS scatter with exact integer sums (no reverse index); see include/dmet.h.
    max_nodes: the batch's largest event when the caller knows it (a hint: workgroups sized for small events).
    sliced: gQ comes back slice-major, [8, N, 4] (for edgeconv_linear_bwd(..., gq_sliced=True))."""
    dev = _require_device(g_out, arg, nbr, ptr)
    L = _lib.load()
    g_out = _f32c(g_out, "g_out")
    N, H = g_out.shape
    if arg.dtype != torch.uint8 or not arg.is_contiguous() or nbr.dtype != torch.int32 or not nbr.is_contiguous():
        raise TypeError("gather_max_bwd_lds: arg must be contiguous uint8, nbr contiguous int32")
    B = ptr.numel() - 1
    sliced = bool(sliced and H == 32)
    gQ = torch.empty((8, N, 4) if sliced else (N, H), dtype=torch.float32, device=dev)
    _t = timer.record('gather_max_bwd', dev)
    with _on(dev):
        if nbr_local is not None and (nbr_local.shape != nbr.shape or nbr_local.dtype != torch.int16
                                      or not nbr_local.is_contiguous()):
            raise ValueError("nbr_local must be the contiguous int16 [N, k] table of knn_local()")
        entry = L.dmet_gather_max_bwd_sliced_f32 if sliced else L.dmet_gather_max_bwd_lds16_cap_f32
        _lib.check(entry(g_out.data_ptr(), arg.data_ptr(), nbr.data_ptr(),
                         nbr_local.data_ptr() if nbr_local is not None else None,
                         ptr.data_ptr(), B, N, nbr.shape[1], H, gQ.data_ptr(),
                         int(max_nodes or 0), _stream(dev)),
                   "dmet_gather_max_bwd_sliced_f32" if sliced else "dmet_gather_max_bwd_lds16_cap_f32")
    if _t is not None:
        _t.record(torch.cuda.current_stream(dev))
    return gQ


def met_loss(met: torch.Tensor, truth: torch.Tensor):
    """(loss[1], d loss / d met [B,2]) of 0.5 * mean((met + truth)^2 summed over px, py)."""
    dev = _require_device(met, truth)
    L = _lib.load()
    met = _f32c(met, "met"); truth = _f32c(truth, "truth")
    if met.dim() != 2 or met.shape[1] != 2 or truth.shape[0] != met.shape[0] or truth.shape[1] < 2:
        raise ValueError("met_loss: met must be [B,2], truth [B,>=2]")
    B = met.shape[0]
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    g = torch.empty_like(met)
    with _on(dev):    # px, py are columns 0, 1 of the rows: no [B,2] copy of the [B,11] target
        _lib.check(L.dmet_met_loss_strided_f32(met.data_ptr(), truth.data_ptr(), truth.stride(0), B, loss.data_ptr(),
                                               g.data_ptr(), _stream(dev)), "dmet_met_loss_strided_f32")
    return loss, g


def _head_params(params, dev):
    shapes = [(16, 32), (16,), (1, 16), (1,)]
    if len(params) != 4:
        raise ValueError("head: expected (W1, b1, W2, b2)")
    out = []
    for t, shp in zip(params, shapes):
        if tuple(t.shape) != shp or t.device != dev:
            raise ValueError(f"head: parameter of shape {tuple(t.shape)}, expected {shp} on the device of emb")
        out.append(_f32c(t, "param"))
    return out


def head_fwd(emb: torch.Tensor, params) -> torch.Tensor:
    """sigmoid(W2 . ELU(W1 . emb + b1) + b2) per node: [N] from emb[N,32]."""
    dev = _require_device(emb)
    L = _lib.load()
    emb = _f32c(emb, "emb")
    if emb.dim() != 2 or emb.shape[1] != 32:
        raise ValueError("head: emb must be [N,32]")
    ps = _head_params(params, dev)
    N = emb.shape[0]
    out = torch.empty((N,), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.dmet_head_fwd_f32(emb.data_ptr(), N, *[t.data_ptr() for t in ps], out.data_ptr(), _stream(dev)),
                   "dmet_head_fwd_f32")
    return out


def bn_head_fwd(raw: torch.Tensor, residual: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                mean: torch.Tensor, invstd: torch.Tensor, params):
    """(emb, out): emb = residual + BatchNorm(raw) (statistics given) formed inside the head's forward launch and out =
    head_fwd(emb, params) (dmet_bn_head_fwd_f32); None when nothing was launched (the caller keeps the two steps)."""
    import ctypes
    dev = _require_device(raw)
    L = _lib.load()
    raw = _f32c(raw.detach(), "raw")
    if raw.dim() != 2 or raw.shape[1] != 32:
        return None
    if residual is not None:
        residual = _f32c(residual.detach(), "residual")
    gamma = _f32c(gamma.detach(), "gamma"); beta = _f32c(beta.detach(), "beta")
    ps = _head_params(params, dev)
    N = raw.shape[0]
    emb = torch.empty_like(raw)
    out = torch.empty((N,), dtype=torch.float32, device=dev)
    fused = ctypes.c_int(0)
    with _on(dev):
        _lib.check(L.dmet_bn_head_fwd_f32(raw.data_ptr(), residual.data_ptr() if residual is not None else None,
                                          gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                          emb.data_ptr(), N, *[t.data_ptr() for t in ps], out.data_ptr(),
                                          ctypes.cast(ctypes.pointer(fused), ctypes.c_void_p), _stream(dev)),
                   "dmet_bn_head_fwd_f32")
    return (emb, out) if fused.value else None


def head_bwd(emb: torch.Tensor, params, out: torch.Tensor, g_out: torch.Tensor):
    """(g_emb, gW1, gb1, gW2, gb2) of head_fwd."""
    dev = _require_device(emb, out, g_out)
    L = _lib.load()
    emb = _f32c(emb, "emb"); out = _f32c(out, "out"); g_out = _f32c(g_out, "g_out")
    ps = _head_params(params, dev)
    N = emb.shape[0]
    g_emb = torch.empty_like(emb)
    grads = [torch.empty_like(t) for t in ps]
    if N == 0:
        return [g_emb] + [g.zero_() for g in grads]
    with _on(dev):
        ws = _ws(L.dmet_head_bwd_workspace_bytes(N), dev)
        _lib.check(L.dmet_head_bwd_f32(emb.data_ptr(), N, ps[0].data_ptr(), ps[1].data_ptr(), ps[2].data_ptr(),
                                       out.data_ptr(), g_out.data_ptr(), g_emb.data_ptr(),
                                       *[g.data_ptr() for g in grads], ws.data_ptr(), ws.numel(), _stream(dev)),
                   "dmet_head_bwd_f32")
        _defer_keep(dev, ws)
    return [g_emb] + grads


def table_rowptr(nbr: torch.Tensor, cnt: Optional[torch.Tensor]) -> torch.Tensor:
    """rowptr[N+1] int32: exclusive prefix sum of the number of valid entries per row of a neighbour table."""
    dev = _require_device(nbr)
    L = _lib.load()
    N, k = nbr.shape
    rowptr = torch.zeros((N + 1,), dtype=torch.int32, device=dev)
    if N:
        deg = torch.empty((N,), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.check(L.dmet_table_degree(nbr.data_ptr(), cnt.data_ptr() if cnt is not None else None, N, k,
                                           deg.data_ptr(), _stream(dev)), "dmet_table_degree")
        torch.cumsum(deg, 0, dtype=torch.int32, out=rowptr[1:])
    return rowptr


def table_edges(nbr: torch.Tensor, cnt: Optional[torch.Tensor], rowptr: torch.Tensor, num_edges: int, swap: bool,
                want_index64: bool, want_int32: bool):
    """(edge_index [2,E] int64 or None, src32 [E] or None, tgt32 [E] or None) of a neighbour table."""
    dev = _require_device(nbr, rowptr)
    L = _lib.load()
    N, k = nbr.shape
    ei = torch.empty((2, num_edges), dtype=torch.int64, device=dev) if want_index64 else None
    s32 = torch.empty((num_edges,), dtype=torch.int32, device=dev) if want_int32 else None
    t32 = torch.empty((num_edges,), dtype=torch.int32, device=dev) if want_int32 else None
    if N and num_edges:
        with _on(dev):
            _lib.check(L.dmet_table_edges(nbr.data_ptr(), cnt.data_ptr() if cnt is not None else None, rowptr.data_ptr(),
                                          N, k, 1 if swap else 0,
                                          ei[0].data_ptr() if ei is not None else None,
                                          ei[1].data_ptr() if ei is not None else None,
                                          s32.data_ptr() if s32 is not None else None,
                                          t32.data_ptr() if t32 is not None else None, _stream(dev)), "dmet_table_edges")
    return ei, s32, t32
