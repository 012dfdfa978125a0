"""What a dependent tiny launch costs on this chip: chains of K tiny kernels (dmet_bn_eval_stats_f32: one 64-thread
workgroup, 32 loads, 64 stores) eager and replayed as a hipGraph, with and without a large streaming kernel in front of
every tiny one (a kernel boundary after 37 MB of writes has the L2 write-back to pay)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
dev = torch.device("cuda:0")
rm, rv = torch.randn(32, device=dev), torch.rand(32, device=dev) + 0.5
big = torch.randn(288000, 32, device=dev)
gamma, beta = torch.ones(32, device=dev), torch.zeros(32, device=dev)
mean, invstd = _native.bn_eval_stats(rm, rv, 1e-5)

def tiny(): _native.bn_eval_stats(rm, rv, 1e-5)
def large(): _native.bn_apply(big, None, gamma, beta, mean, invstd)      # 37 MB read + 37 MB written

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
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
    return g.replay

K = 50
chains = {
    "K tiny": lambda: [tiny() for _ in range(K)],
    "K large": lambda: [large() for _ in range(K)],
    "K x (large, tiny)": lambda: [(large(), tiny()) for _ in range(K)],
    "K x (large, tiny, tiny)": lambda: [(large(), tiny(), tiny()) for _ in range(K)],
}
res = {}
for name, body in chains.items():
    e = timed(body, reps=5)
    r = timed(graphed(body), reps=20)
    res[name] = (e, r)
    print(f"{name:26s} eager {e / K:8.2f} us per element   hipGraph {r / K:8.2f} us per element", flush=True)
lg = res["K large"][1] / K
print(f"a tiny kernel behind a large one costs {res['K x (large, tiny)'][1] / K - lg:.2f} us of graph time; a second one "
      f"{(res['K x (large, tiny, tiny)'][1] - res['K x (large, tiny)'][1]) / K:.2f} us; in a chain of its own {res['K tiny'][1] / K:.2f} us")
