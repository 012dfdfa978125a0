"""A/B of the two kNN paths (matrix-core filter + re-rank vs exact VALU kernel): same bits, timing of each.
Run the process twice (the path is chosen once per process): DMET_KNN_PATH=exact python tools/knn_ab.py ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native

n, D, k = 4500, int(os.environ.get('KNN_AB_D', '32')), 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B * n, D, device=dev)
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
st = {}
nbr, dist = _native.knn(x, ptr, k, stats=st); torch.cuda.synchronize()
print("stats", st)
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _native.knn(x, ptr, k); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print(f"path={os.environ.get('DMET_KNN_PATH', 'filter')} knn {B}x{n}x{D} k={k}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms")
if out:
    if os.path.exists(out):
        ref = torch.load(out)
        print("nbr equal:", torch.equal(ref["nbr"], nbr.cpu()), " dist equal:", torch.equal(ref["dist"], dist.cpu()),
              " mismatching rows:", int((ref["nbr"] != nbr.cpu()).any(dim=1).sum()))
    else:
        torch.save({"nbr": nbr.cpu(), "dist": dist.cpu()}, out)
