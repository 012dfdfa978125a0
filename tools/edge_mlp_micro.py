"""Micro-benchmark of the fused per-edge bf16-MFMA edge MLP (csrc/edgemlp.hip) at BASELINE configs[2] sizes:
64 events x 4500 nodes, k = 16, against the fp32 route for the same nn (edge features -> torch nn -> segment max).
Prints time, useful and executed bf16 flops, and the share of the dense bf16 MFMA peak (2.5 PFLOP/s); run under
`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...` for the counter-based utilisation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import _native

dev = torch.device("cuda:0")
B, n, k = 64, 4500, 16
shapes = [(32, 48, 32), (32, 64, 32), (64, 96, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split(","))]


def med(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for H, H1, H2 in shapes:
    N = B * n
    torch.manual_seed(0)
    x = torch.randn(N, H, device=dev)
    batch = torch.arange(B, device=dev).repeat_interleave(n)
    ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
    dm.register_batch(batch, ptr, B, max_nodes=n)
    nn_ = torch.nn.Sequential(torch.nn.Linear(2 * H, H1), torch.nn.ELU(), torch.nn.Linear(H1, H2), torch.nn.ELU()).to(dev)
    table = dm.knn_table(x, k, batch, loop=True)
    l1, l2 = nn_[0], nn_[2]
    with torch.no_grad():
        t_mfma = med(lambda: _native.edge_mlp2_bf16(x, table.nbr, l1.weight, l1.bias, l2.weight, l2.bias, True, False))
        conv = dm.EdgeConv(nn=nn_)            # re-initialises nn; only timing matters below
        t_f32 = med(lambda: conv(x, table), reps=3)
        # the DRN's nn as written: + BatchNorm1d over the messages (aggr add), one pass + (a, b) + node-level apply
        bn = torch.nn.BatchNorm1d(H2).to(dev)
        t_bn = med(lambda: _native.edge_mlp2_bn_bf16(x, table.nbr, l1.weight, l1.bias, l2.weight, l2.bias, True, True, bn.weight,
                                                     bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                                     bn.num_batches_tracked, True))
        t_add = med(lambda: _native.edge_mlp2_bf16(x, table.nbr, l1.weight, l1.bias, l2.weight, l2.bias, True, True))
        conv_bn = dm.EdgeConv(nn=torch.nn.Sequential(torch.nn.Linear(2 * H, H1), torch.nn.ELU(), torch.nn.Linear(H1, H2),
                                                     torch.nn.ELU(), torch.nn.BatchNorm1d(H2)).to(dev), aggr="add")
        t_f32_bn = med(lambda: conv_bn(x, table), reps=3)
    print(f"H={H} H1={H1} H2={H2}: with trailing BatchNorm1d, aggr add: fused {t_bn * 1e3:8.1f} us (without the norm {t_add * 1e3:8.1f} us); "
          f"fp32 route {t_f32_bn * 1e3:8.1f} us ({t_f32_bn / t_bn:4.1f}x)")
    E = N * k
    useful = 2.0 * E * (2 * H * H1 + H1 * H2)
    H1P = 64 if H == 32 else (96 if H1 <= 96 else 128)
    executed = 2.0 * E * (2 * H * H1P + H1P * H2)
    print(f"H={H} H1={H1} H2={H2} k={k} E={E}: fused bf16 MFMA {t_mfma * 1e3:8.1f} us  "
          f"useful {useful / (t_mfma * 1e-3) / 1e12:6.1f} TFLOP/s, executed {executed / (t_mfma * 1e-3) / 1e12:6.1f} TFLOP/s "
          f"= {executed / (t_mfma * 1e-3) / 2.5e15 * 100:4.1f} % of the dense bf16 MFMA peak;  fp32 route (materialised [E,2H]) "
          f"{t_f32 * 1e3:8.1f} us  ({t_f32 / t_mfma:4.1f}x)")
