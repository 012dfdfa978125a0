"""Host logic on the CPU: the same checks as tests/test_gpu_parity.py, with deepmetv2_amd._native replaced by the
oracle-backed stand-in of tests/fake_native.py (test infrastructure).  Exercises autograd wiring, registries, the
PyG-shaped argument handling and error behaviour without a GPU; the HIP kernels themselves are only tested by -m gpu.
"""
import pytest
import torch

import test_gpu_parity as P
from fake_native import install

CPU = torch.device("cpu")


@pytest.fixture(autouse=True)
def _fake(monkeypatch):
    install(monkeypatch)


@pytest.mark.parametrize("sizes,D,k", [([40, 3, 0, 90], 8, 6), ([1, 2], 3, 4)])
def test_knn(sizes, D, k):
    P.test_knn_bit_exact(CPU, sizes, D, k)


@pytest.mark.parametrize("loop", [True, False])
@pytest.mark.parametrize("flow", ["source_to_target", "target_to_source"])
def test_knn_graph(loop, flow):
    P.test_knn_graph_edge_index(CPU, loop, flow)


def test_radius():
    P.test_radius_graph(CPU)


@pytest.mark.parametrize("sizes,H,k", [([50, 1, 0, 70], 32, 8), ([30, 40], 64, 5)])
def test_fused(sizes, H, k):
    P.test_dynamic_edgeconv_fused_fwd_bwd(CPU, sizes, H, k)


@pytest.mark.parametrize("aggr", ["max", "add", "mean"])
def test_generic(aggr):
    P.test_edgeconv_generic_nn_irregular_graph(CPU, aggr)


def test_static_radius():
    P.test_edgeconv_static_graph_from_radius(CPU)


def test_ties():
    P.test_max_ties_and_empty_rows(CPU)


def test_full_model():
    P.test_full_model_train_step_matches_oracle(CPU)


@pytest.mark.parametrize("ncols", [6, 11])
def test_eval_metrics(ncols):
    P.test_eval_metrics_match_oracle(CPU, ncols)


def test_to_undirected_and_drn_call_shape():
    """The DRN call shape (dynamic_reduction_network.py:86-87): to_undirected(knn_graph(x, k, batch, loop=False)) fed
    to an add-aggregating EdgeConv with a multi-layer nn; to_undirected itself against a set-based restatement."""
    import deepmetv2_amd as dm
    from oracle import ref_ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(60, 5, generator=g)
    batch = torch.repeat_interleave(torch.arange(3), torch.tensor([25, 5, 30]))
    ei = dm.knn_graph(x, 4, batch, loop=False)
    und = dm.to_undirected(ei)
    pairs = sorted({(int(a), int(b)) for a, b in ei.t().tolist()} | {(int(b), int(a)) for a, b in ei.t().tolist()})
    assert und.dtype == torch.int64 and und.t().tolist() == [list(p) for p in pairs]
    nn_ = torch.nn.Sequential(torch.nn.Linear(10, 8), torch.nn.ELU(), torch.nn.Linear(8, 8), torch.nn.ELU())
    conv = dm.EdgeConv(nn=nn_, aggr="add")
    out = conv(x, und)
    ref = ref_ops.edge_conv(x, und, nn_, aggr="add")
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_product_refuses_cpu_tensors():
    """Without the stand-in the product path must fail loudly on CPU tensors (no silent fallback)."""
    import importlib

    import deepmetv2_amd._native as nat
    importlib.reload(nat)
    with pytest.raises(RuntimeError, match="non-GPU tensor"):
        nat.knn(torch.randn(4, 2), torch.tensor([0, 4]), 2)


def test_edgeconv_api_surface():
    import deepmetv2_amd as dm
    lin = torch.nn.Sequential(torch.nn.Linear(64, 32))
    conv = dm.EdgeConv(nn=lin).jittable()
    assert conv.nn is lin and conv.aggr == "max" and conv.flow == "source_to_target"
    assert list(conv.state_dict().keys()) == ["nn.0.weight", "nn.0.bias"]      # owns nothing of its own
    dyn = dm.DynamicEdgeConv(lin, k=16)
    assert dyn.k == 16 and list(dyn.state_dict().keys()) == ["nn.0.weight", "nn.0.bias"]
    with pytest.raises(ValueError):
        dm.EdgeConv(lin, aggr="median")
    with pytest.raises(NotImplementedError):
        dm.knn_graph(torch.zeros(4, 2), 2, cosine=True)
    with pytest.raises(ValueError, match="sorted"):
        dm.knn_graph(torch.zeros(4, 2), 2, torch.tensor([0, 1, 0, 1]))
    with pytest.raises(ValueError):
        dm.knn_graph(torch.zeros(4, 2), 100)


def test_flat_adamw_argument_checks():
    """deepmetv2_amd.optim.FlatAdamW validates like torch.optim.AdamW and refuses non-GPU parameters (no CPU fallback)."""
    import pytest
    import torch
    from deepmetv2_amd.optim import FlatAdamW
    p = torch.nn.Parameter(torch.zeros(8))
    for bad in (dict(lr=-1.0), dict(betas=(1.0, 0.9)), dict(eps=-1e-8), dict(weight_decay=-0.1)):
        with pytest.raises(ValueError):
            FlatAdamW([p], **bad)
    opt = FlatAdamW([p], lr=1e-3)
    opt.step()                      # no gradient yet: nothing to do
    p.grad = torch.ones(8)
    with pytest.raises((TypeError, RuntimeError)):
        opt.step()                  # a CPU parameter: the HIP library is the only implementation


def test_dynamic_edgeconv_rider_requests_follow_the_layer_shape(monkeypatch):
    """Host-side decisions of the third session: which layers ask the kNN build to carry their dense layer / the
    BatchNorm transform before them (cuda, fused fp32 or bf16 form on 32 -> 32 features, k <= 20)."""
    import torch
    import deepmetv2_amd as dm
    from deepmetv2_amd import conv
    x = torch.zeros(10, 32)
    layer = dm.DynamicEdgeConv(torch.nn.Linear(64, 32), k=16)
    assert layer._dense_request(x) is None                       # CPU tensor: nothing native to ask
    assert layer.prebuild_hook(None) is not None
    assert layer.prebuild_hook(None)(x, None, None, None, None, None) is None    # not on the GPU: the caller keeps the transform
    assert dm.DynamicEdgeConv(torch.nn.Linear(64, 32), k=24).prebuild_hook(None) is None    # k > 20: exact kernel
    monkeypatch.setattr(conv, "BN_KNN_FUSE", "0")
    assert layer.prebuild_hook(None) is None
    monkeypatch.setenv("DMET_KNN_PATH", "exact")
    monkeypatch.setattr(conv, "BN_KNN_FUSE", "1")
    assert layer.prebuild_hook(None) is None
    assert layer._take_prebuilt(x) is None


def test_nn_shim_and_accelerate_structure(monkeypatch):
    """deepmetv2_amd.nn re-exports torch.nn with three subclasses; accelerate() swaps classes in place, recognises the
    graph-MET wiring (and only that) and shares parameter objects with the fused Net it returns."""
    import torch
    from tests import fake_native
    fake_native.install(monkeypatch)
    import deepmetv2_amd as dm
    from deepmetv2_amd import nn as dnn, stock_model
    from deepmetv2_amd.model import Net
    assert dnn.ELU is torch.nn.ELU and dnn.Sequential is torch.nn.Sequential and dnn.Module is torch.nn.Module
    for name in ("Linear", "Embedding", "BatchNorm1d"):
        assert issubclass(getattr(dnn, name), getattr(torch.nn, name)) and getattr(dnn, name) is not getattr(torch.nn, name)
    lin = dnn.Linear(5, 3)
    assert torch.equal(lin(torch.ones(2, 5)), torch.nn.functional.linear(torch.ones(2, 5), lin.weight, lin.bias))   # CPU: parent
    bn = dnn.BatchNorm1d(6)                                  # 6 is not a multiple of 4: torch's forward, no recursion
    assert bn(torch.randn(10, 6)).shape == (10, 6)
    stock = stock_model.StockNet(dm, 8, 3, variant="knn_graph", k=4)
    keys = list(stock.state_dict().keys())
    fused = dm.accelerate(stock, graph="dynamic", k=4)
    assert isinstance(fused, Net) and fused.graphnet.graph == "dynamic" and fused.graphnet.k == 4
    assert list(fused.state_dict().keys()) == keys
    assert all(a is b for a, b in zip(fused.parameters(), stock.parameters()))
    assert all(a is b for a, b in zip(fused.buffers(), stock.buffers()))
    assert type(stock.graphnet.encode_all[0]) is dnn.Linear and type(stock.graphnet.embed_pv) is dnn.Embedding
    # auto: EdgeConv blocks mean the caller's static graph, DynamicEdgeConv blocks a kNN graph with the module's k
    assert dm.accelerate(stock_model.StockNet(dm, 8, 3, variant="static")).graphnet.graph == "static"
    dyn = dm.accelerate(stock_model.StockNet(dm, 8, 3, variant="dynamic", k=7))
    assert dyn.graphnet.graph == "dynamic" and dyn.graphnet.k == 7
    # anything that is not wired like the reference's Net comes back as it is (layers swapped)
    other = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    assert dm.accelerate(other) is other and type(other[0]) is dnn.Linear
    with pytest.raises(ValueError):
        dm.accelerate(stock_model.StockNet(dm, 8, 3), graph="sometimes")


def test_flat_module_aligns_every_parameter_to_16_bytes():
    """Odd-sized parameters in the middle of a model must not push the ones behind them off a 16-byte boundary (the
    fused kernels take 16-byte loads of weights; an unaligned view silently selects a slower, differently rounded form)."""
    import torch
    from deepmetv2_amd.parallel import FlatModule

    class Odd(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.arange(3, dtype=torch.float32))          # 3 floats: 12 bytes
            self.b = torch.nn.Parameter(torch.arange(10, dtype=torch.float32).view(2, 5))
            self.c = torch.nn.Parameter(torch.ones(1))
            self.d = torch.nn.Parameter(torch.full((4,), 7.0))

    m = Odd()
    flat = FlatModule(m)
    assert flat.offsets == [0, 4, 16, 20] and flat.numel == 24
    for p in m.parameters():
        assert p.data_ptr() % 16 == 0
    assert torch.equal(m.b.detach(), torch.arange(10, dtype=torch.float32).view(2, 5))
    pad = torch.ones(24, dtype=torch.bool)
    for off, p in zip(flat.offsets, m.parameters()):
        pad[off:off + p.numel()] = False
    assert torch.count_nonzero(flat.flat_param.detach()[pad]) == 0
    # gradients land in the matching slices; the padding stays zero
    (m.a.sum() * 2 + (m.b * 3).sum() + m.c.sum() + m.d.sum()).backward()
    flat.gather_grads()
    assert torch.equal(flat.flat_grad[0:3], torch.full((3,), 2.0)) and torch.equal(flat.flat_grad[4:14], torch.full((10,), 3.0))
    assert torch.count_nonzero(flat.flat_grad[pad]) == 0
    # the reference model: no padding at all
    from deepmetv2_amd.model import Net
    assert FlatModule(Net(8, 3, graph="dynamic", k=16)).numel == 6641
