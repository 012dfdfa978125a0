"""kNN build rate against the event size (64 events, 32 features, k = 16): pairs per second of the whole build.  Events
below 2048 nodes take the first filter form (bf16-split operands, per-key queue), larger ones the second (fp16 tile
records, per-tile hit masks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
D, k, B = 32, 16, 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
for n in [int(v) for v in sys.argv[1:]] or [250, 500, 1000, 1500, 2000, 2047, 2048, 2500, 3000, 3500, 4000, 4500, 6000, 8000]:
    x = torch.randn(B * n, D, device=dev)
    ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
    for _ in range(3): _native.knn_local(x, ptr, k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(11):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.knn_local(x, ptr, k); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"n={n:5d}  N={B * n:7d}  {t:8.1f} us   {B * n * n / t / 1e6:8.2f} T pairs/s   {t / (B * n) * 1e3:7.2f} ns/node", flush=True)
