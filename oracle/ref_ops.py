"""CPU restatement (the ORACLE) of the third-party graph operators on the DeepMETv2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deepmetv2_amd/`` imports this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker / the CPU
baseline, never as the thing that is measured or shipped.

PARITY UNPINNED.  The operators restated here live in torch_cluster / torch_scatter / torch_geometric, which are
neither vendored under /root/reference nor installed in this image (versions unpinned at
/root/reference/README.md:12-17), and the reference has no tests or golden vectors.  The
restatement follows the published semantics of those operators (rules R1-R6) and is anchored
on the reference's call sites, which are cited per function.  Deliberately PyG-shaped and un-fused
(index_select x2 -> cat -> nn -> scatter) so it doubles as "the reference's CPU path" for the timed CPU baseline.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Callable, Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libdmet_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/dmet_oracle.c with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dmet_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        L.dmet_oracle_knn_f32.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p]
        L.dmet_oracle_knn_f32.restype = c.c_int
        L.dmet_oracle_radius_f32.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_float, c.c_int,
                                             c.c_void_p, c.c_void_p]
        L.dmet_oracle_radius_f32.restype = c.c_int
        L.dmet_oracle_met_f64.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_int, c.c_void_p]
        L.dmet_oracle_met_f64.restype = c.c_int
        _lib = L
    return _lib


# ----------------------------------------------------------------------------------------------------------------
# batch vector <-> ptr  (PyG Batch semantics: `batch` sorted, event b owns nodes ptr[b]..ptr[b+1]-1;
# /root/reference/model/data_loader.py:107-110 builds it through the PyG DataLoader collate)
# ----------------------------------------------------------------------------------------------------------------
def batch_to_ptr(batch: Optional[torch.Tensor], n: int, num_events: Optional[int] = None) -> torch.Tensor:
    if batch is None:
        return torch.tensor([0, n], dtype=torch.int64)
    batch = batch.to(torch.int64).cpu()
    if batch.numel() > 1 and bool((batch[1:] < batch[:-1]).any()):
        raise ValueError("batch vector must be sorted")
    B = int(batch.max()) + 1 if batch.numel() else 0
    if num_events is not None:
        B = max(B, num_events)
    counts = torch.bincount(batch, minlength=B)
    return torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])


# ----------------------------------------------------------------------------------------------------------------
# K1  torch_cluster.knn / knn_graph     call sites: model/graph_met_network.py:63,
#                                        model/dynamic_reduction_network.py:86,94
# ----------------------------------------------------------------------------------------------------------------
def knn_table(x: torch.Tensor, ptr: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fixed-width neighbour table nbr[N,k] (int32 global ids, -1 padded) + dist[N,k] via the C oracle (R1, R2)."""
    x = x.detach().to(torch.float32).contiguous().cpu()
    ptr = ptr.to(torch.int64).contiguous().cpu()
    N, D = x.shape
    nbr = torch.full((N, k), -1, dtype=torch.int32)
    dist = torch.empty((N, k), dtype=torch.float32)
    rc = lib().dmet_oracle_knn_f32(x.data_ptr(), ptr.data_ptr(), ptr.numel() - 1, D, k, nbr.data_ptr(),
                                   dist.data_ptr())
    if rc != 0:
        raise RuntimeError(f"dmet_oracle_knn_f32 failed: {rc}")
    return nbr, dist


_libm = None


def _fmaf(a: float, b: float, c: float) -> float:
    global _libm
    if _libm is None:
        _libm = ctypes.CDLL("libm.so.6")
        _libm.fmaf.argtypes = [ctypes.c_float] * 3
        _libm.fmaf.restype = ctypes.c_float
    return _libm.fmaf(a, b, c)


def knn_table_pyloops(x: torch.Tensor, ptr: torch.Tensor, k: int) -> torch.Tensor:
    """Independent slow restatement of R1/R2 in pure-Python loops (libm fmaf) for tiny cases: cross-checks the C."""
    xs = x.detach().to(torch.float32).cpu().numpy()
    N, D = xs.shape
    out = np.full((N, k), -1, dtype=np.int32)
    p = ptr.tolist()
    for b in range(len(p) - 1):
        for i in range(p[b], p[b + 1]):
            cand = []
            for j in range(p[b], p[b + 1]):
                acc = np.float32(0.0)
                for c in range(D):
                    diff = np.float32(xs[j, c] - xs[i, c])
                    acc = np.float32(_fmaf(float(diff), float(diff), float(acc)))
                if acc < np.float32(1e10):
                    cand.append((float(acc), j))
            cand.sort(key=lambda t: (t[0], t[1]))  # lexicographic (d, j) == strict-'>' insertion in ascending j
            for e, (_, j) in enumerate(cand[:k]):
                out[i, e] = j
    return torch.from_numpy(out)


def knn_graph(x: torch.Tensor, k: int, batch: Optional[torch.Tensor] = None, loop: bool = False,
              flow: str = "source_to_target", cosine: bool = False, num_workers: int = 1) -> torch.Tensor:
    """torch_cluster.knn_graph restated.  edge_index[0] = neighbour j (source), [1] = centre i (target) for
    flow='source_to_target' (R5); edges grouped by ascending i, ascending (d, j) inside a group.  loop=False searches
    k+1 and then drops j == i (upstream masks row != col, so a node whose k+1 list does not contain itself keeps
    all k+1 edges)."""
    if cosine:
        raise NotImplementedError("cosine distance is outside the hot path")
    assert flow in ("source_to_target", "target_to_source")
    ptr = batch_to_ptr(batch, x.shape[0])
    kk = k if loop else k + 1
    nbr, _ = knn_table(x, ptr, kk)
    N = x.shape[0]
    tgt = torch.arange(N, dtype=torch.int64).repeat_interleave(kk)
    src = nbr.reshape(-1).to(torch.int64)
    keep = src >= 0
    if not loop:
        keep &= src != tgt
    src, tgt = src[keep], tgt[keep]
    if flow == "source_to_target":
        return torch.stack([src, tgt], 0)
    return torch.stack([tgt, src], 0)


def radius_graph(x: torch.Tensor, r: float, batch: Optional[torch.Tensor] = None, loop: bool = False,
                 max_num_neighbors: int = 32, flow: str = "source_to_target") -> torch.Tensor:
    """torch_cluster.radius_graph restated (call site train.py:48).  loop=False asks for max_num_neighbors+1 and
    drops j == i, as upstream does."""
    assert flow in ("source_to_target", "target_to_source")
    x = x.detach().to(torch.float32).contiguous().cpu()
    ptr = batch_to_ptr(batch, x.shape[0]).contiguous()
    m = max_num_neighbors if loop else max_num_neighbors + 1
    N, D = x.shape
    nbr = torch.empty((N, m), dtype=torch.int32)
    cnt = torch.empty((N,), dtype=torch.int32)
    rc = lib().dmet_oracle_radius_f32(x.data_ptr(), ptr.data_ptr(), ptr.numel() - 1, D, float(r), m,
                                      nbr.data_ptr(), cnt.data_ptr())
    if rc != 0:
        raise RuntimeError(f"dmet_oracle_radius_f32 failed: {rc}")
    tgt = torch.arange(N, dtype=torch.int64).repeat_interleave(m)
    src = nbr.reshape(-1).to(torch.int64)
    keep = src >= 0
    if not loop:
        keep &= src != tgt
    src, tgt = src[keep], tgt[keep]
    return torch.stack([src, tgt], 0) if flow == "source_to_target" else torch.stack([tgt, src], 0)


# ----------------------------------------------------------------------------------------------------------------
# torch_scatter.scatter(reduce='max'|'sum')         call sites: inside EdgeConv (aggr), model/net.py:55-56
# ----------------------------------------------------------------------------------------------------------------
class _ScatterMaxR4(torch.autograd.Function):
    """out[i,c] = max over e with index[e]==i of src[e,c]; empty rows -> 0 (R3); the gradient goes to the single
    winning edge with the LOWEST edge position among ties (R4, torch_scatter CPU rule)."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        E, H = src.shape
        idx = index.view(-1, 1).expand(E, H)
        out = torch.zeros((dim_size, H), dtype=src.dtype)
        out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
        won = src == out.index_select(0, index)
        epos = torch.where(won, torch.arange(E).view(-1, 1).expand(E, H), torch.full((E, H), E))
        arg = torch.full((dim_size, H), E, dtype=torch.int64).scatter_reduce(0, idx, epos, reduce="amin",
                                                                             include_self=True)
        ctx.save_for_backward(arg)
        ctx.E = E
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g_out, _g_arg):
        (arg,) = ctx.saved_tensors
        E = ctx.E
        H = g_out.shape[1]
        g_src = torch.zeros((E + 1, H), dtype=g_out.dtype)
        g_src.scatter_(0, arg, g_out)  # arg == E (empty row) lands in the spill row
        return g_src[:E], None, None


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return _ScatterMaxR4.apply(src, index, dim_size)


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out=None,
                dim_size: Optional[int] = None) -> torch.Tensor:
    """torch_scatter.scatter_add for the call shape at model/net.py:55-56 (1-D src, 1-D index) and for [E,H] rows
    (aggr='add').  fp32, ascending-index accumulation order (index_add_ on CPU)."""
    if src.dim() == 1:
        n = int(index.max()) + 1 if dim_size is None and index.numel() else (dim_size or 0)
        res = torch.zeros(n, dtype=src.dtype) if out is None else out
        return res.index_add(0, index, src) if out is None else res.index_add_(0, index, src)
    assert dim in (0, -2) or src.dim() == 2
    n = int(index.max()) + 1 if dim_size is None and index.numel() else (dim_size or 0)
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return res.index_add(0, index, src)


# ----------------------------------------------------------------------------------------------------------------
# K2+K3  torch_geometric.nn.EdgeConv / DynamicEdgeConv     constructed model/graph_met_network.py:36-38,
#                                                           model/dynamic_reduction_network.py:72-73
# ----------------------------------------------------------------------------------------------------------------
def edge_conv(x: torch.Tensor, edge_index: torch.Tensor, nn: Callable, aggr: str = "max",
              flow: str = "source_to_target", return_arg: bool = False):
    """PyG EdgeConv.forward restated, un-fused: x_i = x[target], x_j = x[source],
    message = nn(cat([x_i, x_j - x_i], -1)), aggregated per target with max (empty -> 0) or add."""
    i_row, j_row = (1, 0) if flow == "source_to_target" else (0, 1)
    tgt = edge_index[i_row]
    src = edge_index[j_row]
    x_i = x.index_select(0, tgt)
    x_j = x.index_select(0, src)
    msg = nn(torch.cat([x_i, x_j - x_i], dim=-1))
    N = x.shape[0]
    if aggr == "max":
        out, arg = scatter_max(msg, tgt, N)
        return (out, arg) if return_arg else out
    if aggr in ("add", "sum"):
        return scatter_add(msg, tgt, dim_size=N)
    if aggr == "mean":
        s = scatter_add(msg, tgt, dim_size=N)
        deg = torch.bincount(tgt, minlength=N).clamp(min=1).to(s.dtype).view(-1, 1)
        return s / deg
    raise ValueError(f"unsupported aggr {aggr!r}")


def dynamic_edge_conv(x: torch.Tensor, batch: Optional[torch.Tensor], nn: Callable, k: int,
                      aggr: str = "max") -> torch.Tensor:
    """PyG DynamicEdgeConv.forward restated: knn(x, x, k, batch, batch).flip(0) (self included, i.e. loop=True),
    then EdgeConv.  The alternative the reference keeps at model/graph_met_network.py:63."""
    ei = knn_graph(x, k, batch, loop=True)
    return edge_conv(x, ei, nn, aggr)


# ----------------------------------------------------------------------------------------------------------------
# K4  per-event MET sums and the loss      model/net.py:49-62
# ----------------------------------------------------------------------------------------------------------------
def met_sums_f64(w: torch.Tensor, x: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    w = w.detach().to(torch.float32).contiguous().cpu()
    x = x.detach().to(torch.float32).contiguous().cpu()
    ptr = ptr.to(torch.int64).contiguous().cpu()
    B = ptr.numel() - 1
    met = torch.empty((B, 2), dtype=torch.float64)
    rc = lib().dmet_oracle_met_f64(w.data_ptr(), x.data_ptr(), x.stride(0), ptr.data_ptr(), B, met.data_ptr())
    if rc != 0:
        raise RuntimeError(f"dmet_oracle_met_f64 failed: {rc}")
    return met


def loss_fn(weights: torch.Tensor, prediction: torch.Tensor, truth: torch.Tensor, batch: torch.Tensor) -> torch.Tensor:
    """model/net.py:49-62 restated line by line (scatter_add -> index_add)."""
    px = prediction[:, 0]
    py = prediction[:, 1]
    true_px = truth[:, 0]
    true_py = truth[:, 1]
    B = truth.shape[0]
    METx = scatter_add(weights * px, batch, dim_size=B)
    METy = scatter_add(weights * py, batch, dim_size=B)
    return 0.5 * ((METx + true_px) ** 2 + (METy + true_py) ** 2).mean()


# ----------------------------------------------------------------------------------------------------------------
# N2  raw-file decoding, event by event           model/data_loader.py:63-90 (METDataset.process)
# ----------------------------------------------------------------------------------------------------------------
def decode_padded_events(x_pad: np.ndarray, y: np.ndarray):
    """Line-by-line restatement of METDataset.process for one npz payload: per event transpose, take columns 3:10,
    insert pX, pY, pT, eta in front, drop rows whose pdgId / charge are the -999 padding, nan_to_num, clip +-5000."""
    out = []
    for ievt in range(np.shape(x_pad)[1]):
        inputs = np.array(x_pad[:, ievt, :]).astype(np.float32).T          # :70-71
        x = inputs[:, 3:10]                                                 # :73
        x = np.insert(x, 0, inputs[:, 0] * np.cos(inputs[:, 2]), axis=1)    # :74
        x = np.insert(x, 1, inputs[:, 0] * np.sin(inputs[:, 2]), axis=1)    # :75
        x = np.insert(x, 2, inputs[:, 0], axis=1)                           # :76
        x = np.insert(x, 3, inputs[:, 1], axis=1)                           # :77
        x = x[x[:, 8] != -999]                                              # :78
        x = x[x[:, 9] != -999]                                              # :79
        x = np.clip(np.nan_to_num(x), -5000.0, 5000.0)                      # :81-82
        out.append((torch.from_numpy(x.astype(np.float32)), torch.from_numpy(np.asarray(y[ievt], np.float32)[None])))
    return out


# ---- evaluation-side metrics (row N4): /root/reference/model/net.py:64-157, restated for the CPU ----------------------
def _recoil(vec: torch.Tensor, v_qt: torch.Tensor):
    """response = v.q / q.q; u_par = |response q| - |q|; u_perp = |v - response q|   (net.py:139-145)"""
    dot = torch.einsum("bi,bi->b", vec, v_qt)
    qq = torch.einsum("bi,bi->b", v_qt, v_qt)
    response = dot / qq
    v_par = torch.einsum("b,bi->bi", response, v_qt)
    u_par = torch.sqrt(torch.einsum("bi,bi->b", v_par, v_par)) - torch.sqrt(qq)
    diff = vec - v_par
    u_perp = torch.sqrt(torch.einsum("bi,bi->b", diff, diff))
    return u_perp, u_par, response


def u_perp_par_loss(weights, prediction, truth, batch):
    """net.py:70-90 (q_T built from truth[:,0] twice, as the reference does)."""
    B = truth.shape[0]
    v_qt = torch.stack((truth[:, 0], truth[:, 0]), dim=1)
    mx = -scatter_add(weights * prediction[:, 0], batch, dim_size=B)
    my = -scatter_add(weights * prediction[:, 1], batch, dim_size=B)
    u_perp, u_par, _ = _recoil(torch.stack((mx, my), dim=1), v_qt)
    return 0.5 * (u_par ** 2 + u_perp ** 2).mean()


def resolution(weights, prediction, truth, batch):
    """net.py:92-157."""
    B = truth.shape[0]
    v_qt = torch.stack((truth[:, 0], truth[:, 1]), dim=1)
    mx = scatter_add(weights * prediction[:, 0], batch, dim_size=B)
    my = scatter_add(weights * prediction[:, 1], batch, dim_size=B)

    def compute(vec):
        return [t.detach().cpu().numpy() for t in _recoil(vec, v_qt)]

    out = {"MET": compute(-torch.stack((mx, my), dim=1)),
           "pfMET": compute(torch.stack((truth[:, 2], truth[:, 3]), dim=1)),
           "puppiMET": compute(torch.stack((truth[:, 4], truth[:, 5]), dim=1))}
    if truth.shape[1] > 6:
        out["deepMETResponse"] = compute(torch.stack((truth[:, 6], truth[:, 7]), dim=1))
        out["deepMETResolution"] = compute(torch.stack((truth[:, 8], truth[:, 9]), dim=1))
    return out, torch.sqrt(truth[:, 0] ** 2 + truth[:, 1] ** 2).detach().cpu().numpy()
