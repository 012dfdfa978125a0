import sys, os
sys.path.insert(0, os.getcwd())
import torch
from deepmetv2_amd import _native, synth
dev = torch.device("cuda:0")
sizes = synth.ragged_sizes(64, 500, 8000, seed=1234)
print("sizes min/max", min(sizes), max(sizes), "n<800:", sum(1 for s in sizes if s < 800))
N = sum(sizes); g = torch.Generator().manual_seed(1)
x = torch.randn(N, 32, generator=g).to(dev)
ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)]).to(dev)
st = {}
_native.knn_local(x, ptr, 16, stats=st); print("gaussian:", st)
# the model's embedding-like data: BatchNorm output scale
x2 = (torch.randn(N, 32, generator=g) * 1.0 + 0.3).to(dev)
st = {}
_native.knn_local(x2, ptr, 16, stats=st); print("shifted:", st)
