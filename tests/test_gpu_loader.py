"""N2 on the device: the reference's `for data in dataloader: data.to(device)` loop (train.py:39-52,
model/data_loader.py:63-110) through this package -- raw padded arrays -> events_from_padded -> EventLoader -> collate
-> DeviceLoader (pinned staging, side-stream copies) -> the HIP training step, with device->host synchronisation
FORBIDDEN while the step runs (torch.cuda.set_sync_debug_mode("error"))."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fake_file(n_evt=10, n_max=900, seed=0):
    """The wire format of data_*/generate_npz.py:125-140: x[12, n_evt, n_max] padded with -999, y[n_evt, 11]."""
    rng = np.random.default_rng(seed)
    x = np.full((12, n_evt, n_max), -999.0, dtype=np.float32)
    sizes = rng.integers(200, n_max + 1, n_evt)
    sizes[1] = n_max
    for e, n in enumerate(sizes):
        x[0, e, :n] = np.clip(rng.exponential(2.0, n), 0.01, 500)
        x[1, e, :n] = rng.uniform(-5, 5, n)
        x[2, e, :n] = rng.uniform(-np.pi, np.pi, n)
        x[3:5, e, :n] = rng.normal(0, 0.1, (2, n))
        x[5, e, :n] = rng.choice([0.0, 0.1396, 0.000511], n)
        x[6, e, :n] = rng.uniform(0, 1, n)
        x[7, e, :n] = rng.choice([211, -211, 130, 22, 11, -13, 1, 2], n)
        x[8, e, :n] = rng.choice([-1, 0, 1], n)
        x[9, e, :n] = rng.integers(0, 4, n)
        x[10:, e, :n] = rng.integers(0, 3, (2, n))
    y = rng.normal(0, 30, (n_evt, 11)).astype(np.float32)
    return x, y, sizes


def test_loader_feeds_training_without_host_sync(dev):
    import deepmetv2_amd as dm
    from deepmetv2_amd.model import Net
    from deepmetv2_amd.parallel import FlatModule, GradSync, train_step
    from oracle import ref_model, ref_ops
    from deepmetv2_amd.model import split_features

    xpad, y, sizes = _fake_file()
    events = dm.events_from_padded(xpad, y)
    assert [e[0].shape[0] for e in events] == sizes.tolist()
    host = dm.EventLoader(events, batch_size=4)                 # ragged batches of 4, 4, 2 events
    assert len(host) == 3

    def run(feed):
        torch.manual_seed(0)
        model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
        flat = FlatModule(model)
        sync = GradSync(flat)
        opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)
        losses = []
        for b in feed():
            assert b.x.device == dev and b.batch.dtype == torch.int64
            torch.cuda.set_sync_debug_mode("error")             # any implicit device->host sync raises from here ...
            try:
                loss = train_step(model, flat, sync, opt, b.x, b.y, b.batch, b.ptr)
            finally:
                torch.cuda.set_sync_debug_mode("default")        # ... to here
            losses.append(loss)
        return [float(l) for l in losses], flat.flat_param.detach().cpu()

    l_pre, p_pre = run(lambda: dm.DeviceLoader(host, dev, depth=2))
    l_plain, p_plain = run(lambda: (b.to(dev) for b in host))
    assert len(l_pre) == 3 and all(np.isfinite(l_pre))
    assert l_pre == l_plain and torch.equal(p_pre, p_plain)    # prefetched copies: same bits as blocking ones

    # first step against the CPU oracle on the same collated batch
    b0 = next(iter(host))
    torch.manual_seed(0)
    ref = ref_model.RefNet(8, 3, graph="dynamic", k=16).train()
    torch.manual_seed(0)
    ref.load_state_dict(Net(8, 3, graph="dynamic", k=16).state_dict())
    w = ref(*split_features(b0.x), None, b0.batch)
    loss_ref = float(ref_ops.loss_fn(w, b0.x, b0.y, b0.batch))
    assert abs(l_pre[0] - loss_ref) <= 1e-4 * abs(loss_ref) + 1e-3
