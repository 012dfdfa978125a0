"""A/B of the LDS gather kernel with int32 global ids vs uint16 event-local ids (experiment build, DMET_GML_MODE=3)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, _lib
B, n, H, k = 64, 4500, 32, 16
dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.randn(B * n, H, device=dev)
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
nbr, _ = _native.knn(x, ptr, k)
W = torch.randn(H, 2 * H, device=dev) * 0.1; b = torch.randn(H, device=dev)
P, Q = _native.node_linear_split(x, W, b)
lo = torch.repeat_interleave(ptr[:-1], n).to(torch.int32)[:, None]
loc = torch.where(nbr < 0, torch.full_like(nbr, 0xFFFF), nbr - lo)
nbr16 = (loc & 0xFFFF).to(torch.int32)
packed = (nbr16[:, 0::2] | (nbr16[:, 1::2] << 16)).contiguous()     # [N, 8] int32 = 16 u16 per row
big = torch.empty(96 << 20, device=dev)   # flush buffer (384 MB > MALL)
def timeit(f, reps=20, flush=False):
    for _ in range(3): f()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if flush: big.add_(1.0)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); e.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(e)
    return tot / reps * 1e3
os.environ["DMET_GML_MODE"] = "0"
ref = _native.gather_max(P, Q, nbr, ptr, True, lds=True)
for flush in (False, True):
    print("flush", flush, "int32 ids:", round(timeit(lambda: _native.gather_max(P, Q, nbr, ptr, True, lds=True), flush=flush), 2), "us")
# u16 path: call the C entry directly (the wrapper checks the table shape)
out = torch.empty_like(P); arg = torch.empty(B * n, H, dtype=torch.uint8, device=dev)
lib = _lib.load()
def call16():
    rc = lib.dmet_gather_max_lds_f32(P.data_ptr(), Q.data_ptr(), packed.data_ptr(), ptr.data_ptr(), B, B * n, k, H,
                                 out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
os.environ["DMET_GML_MODE"] = "3"
for flush in (False, True):
    print("flush", flush, "u16 ids:", round(timeit(call16, flush=flush), 2), "us")
print("equal:", torch.equal(out, ref[0]), torch.equal(arg, ref[1]))
