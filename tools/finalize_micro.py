"""Per-call time of the reduction + finalize pairs of a training step, replayed as hipGraph chains (eager chains are bound
by the launch thread): BatchNorm statistics forward / backward, EdgeConv / encoder / head backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
dev = torch.device("cuda:0")
N, H = 288000, 32
torch.manual_seed(0)
x = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev)
rm, rv = torch.zeros(H, device=dev), torch.ones(H, device=dev)
gamma = torch.ones(H, device=dev)
mean, invstd = _native.bn_stats(x, 1e-5, 0.1, rm, rv)

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

def graphed(body):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            body()
    return gr.replay

K = 40
cases = {
    "bn_stats (reduce<0> + finalize)": lambda: _native.bn_stats(x, 1e-5, 0.1, rm, rv),
    "bn_bwd (reduce<1> + finalize + apply)": lambda: _native.bn_bwd(x, g, gamma, mean, invstd),
    "bn_apply alone": lambda: _native.bn_apply(x, None, gamma, gamma, mean, invstd),
}
for name, f in cases.items():
    t = timed(graphed(lambda: [f() for _ in range(K)])) / K
    print(f"{name:42s} {t:7.2f} us per call", flush=True)
