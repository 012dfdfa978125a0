import sys, os
sys.path.insert(0, os.getcwd())
import torch
from deepmetv2_amd import _native
D, k = 32, 16
dev = torch.device("cuda:0"); torch.manual_seed(0)
for n, B in ((4500, 64), (8000, 36), (2000, 128)):
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
    for nbad in (0, 1, 4):
        xx = x.clone()
        idx = torch.randperm(B * n, device=dev)[:nbad]
        xx[idx, 3] = 3.0e4
        t, st = med(xx)
        print(f"n={n} B={B}: {nbad} wide rows: build {t:7.1f} us  flagged {st['flagged_queries']}", flush=True)
