"""The data-parallel step over RCCL (torch.distributed backend "nccl" on ROCm), on the one GPU a test box has:
a process group of ONE rank created in a fresh child process (started before anything touches the GPU), running the
same `train_step` / `GraphedTrainStep` code and the same `all_reduce` call as an N-rank job.  Multi-rank sharding and
gradient averaging are covered on the CPU by tests/test_ddp_gloo.py (gloo, world_size 2)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_rccl_world1_train_step_and_graphed_step(dev):
    env = dict(os.environ)
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_child.py")], env=env, capture_output=True,
                       text=True, timeout=540)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_CHILD ")]
    assert line, p.stdout[-2000:]
    r = json.loads(line[-1][len("RCCL_CHILD "):])
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["nogroup_eager"]["active"] is False and r["rccl_eager"]["active"] is True
    assert r["allreduce_ok"]
    # 3 eager steps through the real all_reduce: bit for bit the run without a process group
    assert r["eager_bitwise_equal"], (r["nogroup_eager"]["losses"], r["rccl_eager"]["losses"])
    assert r["nogroup_eager"]["losses"] == r["rccl_eager"]["losses"]
    # 3 hipGraph replays around the eager all_reduce: bit for bit the eager continuation from the same state
    assert r["graphed_bitwise_equal"], (r["rccl_graphed"], r["rccl_graphed_vs_eager_losses"])
    assert r["rccl_graphed"]["losses"] == r["rccl_graphed_vs_eager_losses"]
    # and no pathology: a replayed step is not slower than ~3x the eager one (small batch: launch-bound either way)
    assert r["rccl_graphed"]["ms"] < 3.0 * r["rccl_eager"]["ms"] + 5.0, r
