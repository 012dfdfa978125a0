"""node_linear_split (the per-node dense layer of the split EdgeConv) standalone: median time at 64 x 4500 x 32 -> 2 x 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
dev = torch.device("cuda:0")
N = 64 * 4500
x = torch.randn(N, 32, device=dev); W = torch.randn(32, 64, device=dev) / 8; b = torch.randn(32, device=dev)
for sliced in (True, False):
    _native.node_linear_split(x, W, b, sliced=sliced); torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.node_linear_split(x, W, b, sliced=sliced); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    ts.sort()
    print(f"node_linear_split sliced={sliced}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us  (110.6 MB: {110.6e6 / (ts[len(ts)//2] * 1e-6) / 1e12:.2f} TB/s)")
