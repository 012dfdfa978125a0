"""What ONE uncertified query costs a kNN build (64 x 4500 x 32, k = 16): rows with a feature beyond the fp16 operand range
are refused as queries by the matrix-core filter and recomputed by the per-query fallback (same bits, include/dmet.h)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
n, D, k, B = 4500, 32, 16, 64
dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.randn(B * n, D, device=dev)
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
def med(xx, reps=15):
    st = {}
    _native.knn_local(xx, ptr, k, stats=st); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.knn_local(xx, ptr, k); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2], st
for nbad in (0, 1, 2, 8, 64, 512):
    xx = x.clone()
    idx = torch.randperm(B * n, device=dev)[:nbad]
    xx[idx, 3] = 3.0e4          # outside the fp16 operand range: refused as a query (and a forced candidate of its event)
    t, st = med(xx)
    print(f"{nbad:4d} wide rows: build {t:7.1f} us   flagged queries {st['flagged_queries']}, tiles {st['flagged_tiles']}", flush=True)
