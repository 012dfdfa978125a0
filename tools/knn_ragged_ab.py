"""kNN build on the configs[4] batch shape (64 ragged events of 500-8000 nodes): median time, fallback counters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 21
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500, 8000)
dev = torch.device("cuda:0")
sizes = synth.ragged_sizes(64, lo, hi, seed=0)
N = sum(sizes)
torch.manual_seed(0)
x = torch.randn(N, 32, device=dev)
ptr = torch.tensor([0] + sizes, dtype=torch.int64).cumsum(0).to(dev)
st = {}
_native.knn(x, ptr, 16, stats=st); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _native.knn(x, ptr, 16); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
pairs = sum(n * n for n in sizes)
print(f"knn ragged {lo}-{hi} x64 (N={N}, {pairs / 1e9:.2f} G pairs, {sum(n < 2048 for n in sizes)} events < 2048): "
      f"median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  {st}")
