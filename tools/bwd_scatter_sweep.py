"""gather_max_bwd_lds_kernel over event sizes: uniform batches of ~288 000 nodes at several event sizes, and the
seeded ragged draw of bench.py --ragged 500 8000 (where is the time of a ragged batch: per-item cost or balance?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth
dev = torch.device("cuda:0"); torch.manual_seed(0)
H = 32
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
def run(sizes, label):
    N = sum(sizes)
    ptr = torch.tensor([0] + sizes, dtype=torch.int64).cumsum(0).to(dev)
    # neighbours: random nodes of the same event (the scatter pattern, not the graph, is what is timed)
    lo = torch.repeat_interleave(ptr[:-1], torch.tensor(sizes, device=dev))
    cnt = torch.repeat_interleave(torch.tensor(sizes, device=dev), torch.tensor(sizes, device=dev))
    nbr = (lo[:, None] + (torch.rand(N, 16, device=dev) * cnt[:, None]).long()).to(torch.int32)
    nbr = torch.minimum(nbr, (lo + cnt - 1)[:, None].to(torch.int32))
    loc = (nbr - lo[:, None].to(torch.int32)).to(torch.int16) if max(sizes) <= 32767 else None
    g = torch.randn(N, H, device=dev)
    arg = torch.randint(0, 16, (N, H), device=dev, dtype=torch.uint8)
    t = timeit(lambda: _native.gather_max_bwd_lds(g, arg, nbr, ptr, nbr_local=loc))
    print(f"{label:28s} B={len(sizes):4d} N={N:7d} {t:7.1f} us  ({t / N * 1e3:.3f} ns/node)", flush=True)
for n in (500, 1000, 2000, 4000, 4500, 4608, 4700, 5200, 6000, 8000, 9216, 9300):
    B = max(1, round(288000 / n))
    run([n] * B, f"uniform n={n}")
run([4500] * 64, "uniform 64 x 4500")
run([8000] * 64, "64 x 8000")
run([8000] * 32, "32 x 8000")
run(synth.ragged_sizes(64, 500, 8000, seed=1234), "ragged U[500,8000]")
run(sorted(synth.ragged_sizes(64, 500, 8000, seed=1234), reverse=True), "ragged, largest first")
run(synth.ragged_sizes(64, 500, 4600, seed=1234), "ragged U[500,4600]")
