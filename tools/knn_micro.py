"""Micro-benchmark of the kNN kernel alone (BASELINE configs[1] shape) -- for rocprofv3 / PMC runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native

n, D, k = 4500, 32, 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B * n, D, device=dev)
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
_native.knn(x, ptr, k); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _native.knn(x, ptr, k); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
flops = 3.0 * B * n * n * D
print(f"knn {B}x{n}x{D} k={k}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms  -> {flops/ts[len(ts)//2]/1e9:.1f} TFLOP/s (3 flop/elem)")
