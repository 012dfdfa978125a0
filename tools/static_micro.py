"""Stage times of the static (radius) flow's graph side at the bench shape: radius table, packing, counted gathers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import _native, synth

dev = torch.device("cuda:0")
B, n = 64, 4500
x, y, batch, ptr = synth.make_events([n] * B, seed=0, device=dev)
dm.register_batch(batch, ptr, B, max_nodes=n)
etaphi = torch.stack([x[:, 3], torch.atan2(x[:, 1], x[:, 0])], 1).contiguous()
H = 32
xe = torch.randn(B * n, H, device=dev)
W = torch.randn(H, 2 * H, device=dev) / 8
bias = torch.randn(H, device=dev)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


t = dm.radius_table(etaphi, 0.4, batch, loop=True, max_num_neighbors=255)
print("mean cnt", float(t.cnt.float().mean()), "max", int(t.cnt.max()))
print("radius_table      us med/min/max", timed(lambda: dm.radius_table(etaphi, 0.4, batch, loop=True, max_num_neighbors=255)))
print("order_by_count    us", timed(lambda: _native.table_order_by_count(t.cnt, t.ptr)))
P, Q = _native.node_linear_split(xe, W, bias, sliced=True)
order = _native.table_order_by_count(t.cnt, t.ptr)
print("gather j16        us", timed(lambda: _native.gather_max_counted_j16(P, Q, t.nbr, t.cnt, order, t.ptr, True)))
print("gather rows16 j16 us", timed(lambda: _native.gather_max_local_j16(P, Q, t.rows16, t.cnt, order, t.ptr, t.k, True)))
