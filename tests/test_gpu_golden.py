"""GPU: the HIP path against the committed golden fixtures (tests/golden/, oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _g(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_g1_config1(dev, golden_dir):
    """BASELINE configs[0]: 1 event x 256 nodes, k=8, one EdgeConv layer (seeded and trained weights)."""
    import deepmetv2_amd as dm
    G = _g(golden_dir, "g1_config1.npz")
    t = dm.knn_table(G["x"].to(dev), 8, None, loop=True)
    assert torch.equal(t.nbr.cpu(), G["nbr"]) and torch.equal(t.dist.cpu(), G["dist"])
    for tag in ("seeded", "trained"):
        lin = torch.nn.Sequential(torch.nn.Linear(64, 32))
        conv = dm.DynamicEdgeConv(lin, k=8)
        with torch.no_grad():
            lin[0].weight.copy_(G[f"W_{tag}"]); lin[0].bias.copy_(G[f"b_{tag}"])
        conv.to(dev)
        out = conv(G["x"].to(dev))
        torch.testing.assert_close(out.detach().cpu(), G[f"out_{tag}"], rtol=1e-5, atol=1e-5)


def test_g2_ties_and_short_events(dev, golden_dir):
    import deepmetv2_amd as dm
    G = _g(golden_dir, "g2_ties.npz")
    t = dm.knn_table(G["x"].to(dev), 16, G["batch"].to(dev), loop=True, num_events=7)
    assert torch.equal(t.nbr.cpu(), G["nbr16"]) and torch.equal(t.dist.cpu(), G["dist16"])
    ei = dm.knn_graph(G["x"].to(dev), 4, G["batch"].to(dev), loop=False)
    assert torch.equal(ei.cpu(), G["ei_k4_noloop"])


def test_g3_ragged(dev, golden_dir):
    import deepmetv2_amd as dm
    G = _g(golden_dir, "g3_ragged.npz")
    lin = torch.nn.Sequential(torch.nn.Linear(64, 32))
    conv = dm.DynamicEdgeConv(lin, k=16)
    with torch.no_grad():
        lin[0].weight.copy_(G["W"]); lin[0].bias.copy_(G["b"])
    conv.to(dev)
    out = conv(G["x"].to(dev), G["batch"].to(dev))
    torch.testing.assert_close(out.detach().cpu(), G["out"], rtol=1e-5, atol=1e-5)
    assert torch.equal(dm.knn_table(G["x"].to(dev), 16, G["batch"].to(dev)).nbr.cpu(), G["nbr"])


def test_g5_irregular(dev, golden_dir):
    import deepmetv2_amd as dm
    G = _g(golden_dir, "g5_irregular.npz")
    nn_ = torch.nn.Sequential(torch.nn.Linear(32, 24), torch.nn.ELU(), torch.nn.Linear(24, 16), torch.nn.ELU())
    for aggr in ("max", "add"):
        conv = dm.EdgeConv(nn_, aggr=aggr)
        nn_.load_state_dict({k.replace("_", ".", 1): v for k, v in G.items() if k[0].isdigit()})
        conv.to(dev)
        out = conv(G["x"].to(dev), G["edge_index"].to(dev))
        torch.testing.assert_close(out.detach().cpu(), G[f"out_{aggr}"], rtol=1e-4, atol=1e-5)
        conv.cpu()


def test_g6_radius(dev, golden_dir):
    import deepmetv2_amd as dm
    G = _g(golden_dir, "g6_radius.npz")
    ei = dm.radius_graph(G["etaphi"].to(dev), 0.4, G["batch"].to(dev), loop=True, max_num_neighbors=255)
    assert torch.equal(ei.cpu().int(), G["ei_r04_loop_255"])
    ei = dm.radius_graph(G["etaphi"].to(dev), 0.4, G["batch"].to(dev), loop=False, max_num_neighbors=12)
    assert torch.equal(ei.cpu().int(), G["ei_r04_noloop_12"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_g4_reference_model_run(dev, golden_dir, mode):
    """The reference's own model code + shipped checkpoint produced G4 (static radius graph, train.py:42-50):
    the product (radius_graph -> EdgeConv x2 -> sigmoid -> MET loss, all HIP) must reproduce it."""
    import deepmetv2_amd as dm
    from deepmetv2_amd.model import Net, loss_fn, split_features
    G = _g(golden_dir, "g4_reference_model.npz")
    sd = {k.replace("__", "."): v for k, v in _g(golden_dir, "g4_checkpoint_dytt_best.npz").items()}
    model = Net(8, 3, graph="static")
    model.load_state_dict(sd)
    model.to(dev)
    getattr(model, mode)()
    x, y, batch = G["x"].to(dev), G["y"].to(dev), G["batch"].to(dev)
    phi = torch.atan2(x[:, 1], x[:, 0])
    etaphi = torch.cat([x[:, 3][:, None], phi[:, None]], dim=1)
    ei = dm.radius_graph(etaphi, r=0.4, batch=batch, loop=True, max_num_neighbors=255)
    # atan2 on the GPU may differ from the CPU's by an ulp, which can flip a borderline dR < 0.4: compare the
    # graph first and fall back to the fixture's graph if (and only if) it differs in a handful of edges
    ref_ei = G["edge_index"].long()
    if ei.shape != ref_ei.shape or not torch.equal(ei.cpu(), ref_ei):
        # every edge present in only one of the two graphs must be borderline: |dR^2 - r^2| within a few ulp of the
        # coordinates' rounding (phi from atan2), never a genuinely different neighbour
        N = x.shape[0]
        a = set((ei[0].cpu() * N + ei[1].cpu()).tolist())
        b = set((ref_ei[0] * N + ref_ei[1]).tolist())
        odd = sorted(a ^ b)
        assert 0 < len(odd) <= 4, len(odd)
        ep = etaphi.double().cpu()
        for key in odd:
            j, i = key // N, key % N
            d2 = float(((ep[j] - ep[i]) ** 2).sum())
            assert abs(d2 - 0.16) < 1e-5, (j, i, d2)
        ei = ref_ei.to(dev)
    w = model(*split_features(x), ei, batch)
    loss = loss_fn(w, x, y, batch)
    loss.backward()
    torch.testing.assert_close(w.detach().cpu(), G[f"weights_{mode}"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss.detach().cpu(), G[f"loss_{mode}"], rtol=1e-4, atol=1e-3)
    g = model.graphnet.conv_continuous[0][0].nn[0].weight.grad.cpu()
    gr = G[f"grad_conv0_weight_{mode}"]
    torch.testing.assert_close(g, gr, rtol=2e-3, atol=2e-4 * float(gr.abs().max()))
