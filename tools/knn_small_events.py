"""kNN build time for batches of small / medium events, with and without a split tail (few events = every tile is a tail
tile): (n, B) pairs around the boundary between the two filter forms."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmetv2_amd import _native
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [(700, 64), (1000, 64), (1000, 32), (1000, 256), (1088, 64), (1100, 64), (1100, 30), (1200, 64), (1400, 64), (1408, 32),
         (1500, 64), (1500, 32), (1500, 20), (2000, 64), (2000, 32), (2040, 16), (2048, 32), (2048, 16), (3000, 16), (4500, 64)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for n, B in cases:
    x = torch.randn(B * n, 32, device=dev)
    ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
    st = {}
    _native.knn(x, ptr, 16, stats=st)
    for _ in range(3): _native.knn_local(x, ptr, 16)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.knn_local(x, ptr, 16); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"n={n:5d} B={B:3d} tiles={B * ((n + 63) // 64):5d} flagged={st['flagged_queries']:5d}  {ts[4]:7.1f} us", flush=True)
