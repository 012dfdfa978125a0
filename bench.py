#!/usr/bin/env python3
"""bench.py -- events/s of the DeepMETv2 DynamicEdgeConv hot path on MI355X (BASELINE.json metric).

A "step" = one full training step (the sequence of /root/reference/train.py:40-52: zero_grad, feature split, model
forward with a kNN graph rebuilt in the embedding before each of the 2 EdgeConv layers, loss, backward, gradient
all-reduce when N>1, AdamW step) over one batch of synthetic events resident in HBM.  Workload at N=1 is
BASELINE.json configs[1]: 64 events x 4500 PF candidates x 11 features, k=16, fp32; for N>1 every rank gets its own
64 events (weak scaling, batch = 64*N events as in configs[3]).

Prints ONE JSON line on rank 0.  Launch for N>1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
FP32_VALU_PEAK_TFLOPS = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--events-per-gpu", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=4500)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = BASELINE configs[2]: edge-MLP dense layer on the bf16 matrix cores, bf16 Q table "
                         "(kNN, max, MET and the backward stay fp32)")
    ap.add_argument("--graph", choices=["dynamic", "static", "static-table"], default="dynamic",
                    help="dynamic = kNN in the embedding before each EdgeConv (north star); static = the active reference "
                         "flow (train.py:42-50): one radius graph dR<0.4 in (eta,phi) per batch, rebuilt every step; "
                         "static-table = the same graph handed to the model as dm.radius_table(...) instead of "
                         "radius_graph's [2,E] tensor (no host sync for the edge count)")
    ap.add_argument("--ragged", type=int, nargs=2, metavar=("LO", "HI"), default=None,
                    help="BASELINE configs[4]: event sizes drawn uniformly from [LO, HI] (seeded) instead of --nodes")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--hip-graph", dest="hip_graph", action="store_true", default=None,
                    help="replay the step as two hipGraphs around the gradient all-reduce (per-kernel HIP-event figures "
                         "then come from a few eager steps after the timed region).  Default: on for --gpus N > 1 (the "
                         "eager step costs the launch thread 0.6-0.9 of the GPU time: N ranks on one host plus the RCCL "
                         "proxy threads have no headroom for it); at N = 1 decided by a probe: on when the launch thread "
                         "needs more than half of the step")
    ap.add_argument("--no-hip-graph", dest="hip_graph", action="store_false")
    ap.add_argument("--model", choices=["fused", "stock-knn-graph", "stock-dynamic"], default="fused",
                    help="fused: this repo's model.Net (fused encoder / head kernels, BatchNorm riders, FlatAdamW); "
                         "stock-*: deepmetv2_amd/stock_model.py -- stock torch.nn layers around ONLY the public operators, "
                         "torch.optim.AdamW on model.parameters(): what the reference's training loop gets with three "
                         "import lines changed (stock-knn-graph: EdgeConv over knn_graph(...), graph_met_network.py:63; "
                         "stock-dynamic: DynamicEdgeConv; with --graph static: EdgeConv over radius_graph, :65)")
    ap.add_argument("--optimizer", choices=["dmet", "torch"], default="dmet",
                    help="AdamW as one HIP launch on the flat parameter tensor (default) or torch.optim.AdamW(fused=True)")
    ap.add_argument("--input", choices=["device", "host"], default="device",
                    help="device: the batch is resident in HBM when the timed region starts (the metric's definition); "
                         "host: every step takes its batch from pinned host memory through deepmetv2_amd.DeviceLoader "
                         "(copied on a side stream two batches ahead), i.e. the PCIe-inclusive rate")
    ap.add_argument("--accelerate", choices=["none", "layers", "fuse"], default="none",
                    help="--model stock-* only: what ONE extra line in the training script buys -- layers: "
                         "deepmetv2_amd.accelerate(model, fuse=False) (torch.nn.Linear / Embedding / BatchNorm1d become the "
                         "HIP-backed subclasses of deepmetv2_amd.nn, the same as a fourth import line `import "
                         "deepmetv2_amd.nn as nn`); fuse: model = deepmetv2_amd.accelerate(model): the graph-MET wiring is "
                         "recognised and replaced by this repo's fused Net sharing the parameters; the loop, the loss with its "
                         "two scatter_add calls and torch.optim.AdamW stay the reference's")
    ap.add_argument("--no-graph-async", dest="graph_async", action="store_false", default=True,
                    help="--graph static-table: build the radius table on the caller's stream (as train.py:48-49 orders it) "
                         "instead of on a side stream beside the encoder")
    ap.add_argument("--prewarm-ms", type=float, default=200.0,
                    help="untimed run-in before the W warm-up steps: the same step repeated for this long, so that the "
                         "module loads, the allocator's growth and the GPU's clock ramp are over when warm-up starts "
                         "(a cold start makes the first ~30 steps 5 %% slower); 0 disables it")
    ap.add_argument("--bracket-every", type=int, default=4,
                    help="the graded kernel's HIP-event bracket is recorded in every n-th step of the timed region "
                         "(a bracket is two extra stream commands)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-events", type=int, default=0, help="0 = one event per host core (max 16)")
    return ap.parse_args()


class _StdoutToStderr:
    """RCCL prints a five-line version banner with printf on the first communicator of a process.  The contract is ONE JSON
    line on stdout: while the communicator is created, file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def usable_cores() -> int:
    """Host cores this process may really use: min(affinity mask, cgroup cpu.max quota, cpu_count)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def sample_sizes(args, n_ev: int):
    """Event sizes of a bounded CPU sample of the benchmarked workload (same distribution: fixed --nodes, or the first
    n_ev sizes of the seeded --ragged draw)."""
    if args.ragged is None:
        return [args.nodes] * n_ev
    from deepmetv2_amd import synth
    return synth.ragged_sizes(max(n_ev, 1), args.ragged[0], args.ragged[1], seed=1234)[:n_ev]


def size_label(args, sizes):
    return str(args.nodes) if args.ragged is None else f"U[{args.ragged[0]},{args.ragged[1]}] ({min(sizes)}..{max(sizes)})"


def cpu_baseline(args, seed: int):
    """The same training step on the host cores through the CPU oracle (PyG-shaped, un-fused restatement of the
    reference's operators; kind 'port' -- PyG itself is not installable here), on a bounded sample of events."""
    import torch

    from deepmetv2_amd import synth
    from oracle import ref_model, ref_ops

    cores = usable_cores()
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    n_ev = args.cpu_sample_events or max(2, min(16, cores))
    sizes = sample_sizes(args, n_ev)
    x, y, batch, ptr = synth.make_events(sizes, seed=seed)
    torch.manual_seed(0)
    model = ref_model.RefNet(8, 3, graph="dynamic", k=args.k).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    def step():
        opt.zero_grad()
        w = model(x[:, :8], x[:, 8:].long(), None, batch)
        loss = ref_ops.loss_fn(w, x, y, batch)
        if args.mode == "train":
            loss.backward()
            opt.step()
        return float(loss.detach())

    step()  # warm-up (builds the C oracle, pages in)
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or (time.perf_counter() - t0 < 8.0 and reps < 10):
        step()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": n_ev / dt, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": f"{n_ev} events x {size_label(args, sizes)} nodes, k={args.k}, 2 layers, {args.mode} step, {reps} reps "
                      f"({dt:.2f} s/step); oracle/ref_model.py + C kNN (OpenMP over events) on {cores} threads"}


def parity_sample(args, model, dev, seed: int):
    """The second half of BASELINE.json's metric, on a bounded sample: MET px/py MSE of the HIP path against the CPU
    oracle (same weights, same seeded events, train-mode BatchNorm statistics) and the number of kNN index mismatches
    of the first layer's graph against the C oracle (must be 0)."""
    import torch

    from deepmetv2_amd import _native, synth
    from deepmetv2_amd.model import split_features
    from deepmetv2_amd.scatter import met_reduce
    from oracle import ref_model, ref_ops

    n_ev = 4
    sizes = sample_sizes(args, n_ev)
    x, y, batch, ptr = synth.make_events(sizes, seed=seed)
    ref = ref_model.RefNet(8, 3, graph="dynamic", k=args.k)
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    ref.train()
    was_training = model.training
    model.train()
    saved = {k: v.detach().clone() for k, v in model.state_dict().items()}   # running statistics are touched below
    with torch.no_grad():
        xd, bd, pd = x.to(dev), batch.to(dev), ptr.to(dev)
        w = model(*split_features(xd), None, bd)
        met = met_reduce(w, xd, ptr=pd).cpu()
        emb = model.graphnet.embed(*split_features(xd)).contiguous()
        nbr, _ = _native.knn(emb, pd, args.k)
        w_ref = ref(*split_features(x), None, batch)
        met_ref = ref_ops.met_sums_f64(w_ref, x, ptr)          # float64 sums of the oracle's weights
        nbr_ref, _ = ref_ops.knn_table(emb.cpu(), ptr, args.k)
    model.load_state_dict(saved)
    model.train(was_training)
    d = met.double() - met_ref.double()
    scale = float((w_ref.double().abs() * x[:, :2].double().abs().sum(1)).sum() / n_ev)
    return {"sample": f"{n_ev} events x {size_label(args, sizes)} nodes, k={args.k}, full 2-layer forward, same weights",
            "met_mse_vs_ref": float((d * d).sum(1).mean() / 2.0),
            "met_max_abs_diff": float(d.abs().max()), "met_sum_abs_wp_per_event": scale,
            "knn_index_mismatches": int((nbr.cpu() != nbr_ref).sum())}


def workload_label(args, B, n, k) -> str:
    which = "configs[1]"
    if args.ragged is not None:
        which = "configs[4] (ragged events)"
    elif args.dtype == "bf16":
        which = "configs[2] (bf16 edge-MLP on MFMA)"
    nodes = f"{n}" if args.ragged is None else f"U[{args.ragged[0]},{args.ragged[1]}]"
    if args.graph == "dynamic":
        graph = "2 DynamicEdgeConv layers (kNN rebuilt per layer in the 32-d embedding)"
    else:
        graph = ("2 EdgeConv layers over one radius graph dR<0.4 in (eta,phi) rebuilt every step (the reference's active "
                 "flow, train.py:48)" + (", table handed to the model" if args.graph == "static-table" else ""))
        which = "reference's active flow at the " + which + " sizes"
    return (f"BASELINE {which}: {B} events/GPU x {nodes} PF candidates x 11 features, k={k}, {graph}, "
            f"{'fp32' if args.dtype == 'f32' else 'bf16 dense layer, fp32 elsewhere'}, {args.mode} step")


def knn_floor(sizes):
    """Stated bound for the matrix-core kNN build (D = 32): per 64-query x 32-candidate tile a wavefront issues 4
    v_mfma_f32_32x32x16_f16 (single-term fp16 operands since round 2's second session; 32 cycles of its SIMD's matrix
    pipe each) and ~130 vector instructions for the selection (half-wave swap, hit mask, tile minimum, threshold
    list) at 4 cycles each (a 64-lane instruction on a 16-lane SIMD: SQ_ACTIVE_INST_VALU x 4 is what the counters
    charge; earlier lines of this file priced them at 2); the first 32 tiles of a sweep are visited twice.  Floors at
    1024 SIMDs x 2.4 GHz if either pipe were the only limit; the vector floor is the one that binds."""
    tiles = 0
    for nn_ in sizes:
        ct = (nn_ + 31) // 32
        tiles += ((nn_ + 63) // 64) * (ct + min(ct, 32))
    clk, simds = 2.4e9, 1024
    return {"tiles": tiles, "mfma_floor_us": round(tiles * 4 * 32 / simds / clk * 1e6, 1),
            "valu_floor_us": round(tiles * 130 * 4 / simds / clk * 1e6, 1)}


def gather_roofline(args, ksum, ev_overhead_ms, model, x, batch, ptr, sizes, static_graph, dev):
    """`roofline` block of the north-star kernel (fused gather + max): algorithmic bytes / mean launch time inside the
    timed region, the name of the kernel form that really ran, and two untimed legs that bracket the cache state --
    the same launch standalone back to back (operands L2 / Infinity-Cache warm) and with 512 MB written between
    launches (cold: everything comes from HBM).  Inside the training step the kernel sits in between: P and Q were
    written by the launch just before, the ids and the rest are cold."""
    import torch

    import deepmetv2_amd as dm
    from deepmetv2_amd import _native
    from deepmetv2_amd.model import split_features

    gname = "edgeconv_fused" if "edgeconv_fused" in ksum else ("gather_max" if "gather_max" in ksum else None)
    if gname is None:
        return None
    H, k, N = 32, args.k, x.shape[0]
    ms = ksum[gname][1]
    g = model.graphnet
    conv = g.conv_continuous[0][0]
    lin = conv.nn[0]
    # algorithmic bytes per launch (SURVEY 8d / BASELINE.md 2): own row H*4 + neighbour ids + output row H*4
    # [+ arg: one byte per channel when training].  ids: k*4 per node for the fixed-k kNN table (int32, the API's
    # internal width; the LDS kernel reads a uint16 copy, not credited), cnt_i*4 for a counted radius table.
    with torch.no_grad():
        emb = (g.embed if hasattr(g, "embed") else g.encode)(*split_features(x)).contiguous()
        if args.graph == "dynamic":
            table = dm.knn_table(emb, k, batch, loop=True)
            id_bytes = float(N) * k * 4
            id_note = f"{k} int32 ids per node"
        else:
            table = static_graph() if args.graph == "static-table" else None
            if isinstance(table, dm.GraphFuture):
                table = table.result()
            if table is None:
                phi = torch.atan2(x[:, 1], x[:, 0])
                table = dm.radius_table(torch.stack([x[:, 3], phi], 1), r=0.4, batch=batch, loop=True, max_num_neighbors=255)
            mean_cnt = float(table.cnt.double().mean())
            id_bytes = float(table.cnt.double().sum()) * 4
            id_note = f"counted radius table, mean {mean_cnt:.1f} int32 ids per node"
    arg_bytes = H if args.mode == "train" else 0
    alg_bytes = N * (H * 4 + H * 4 + arg_bytes) + id_bytes
    ach = alg_bytes / (ms * 1e-3) / 1e9
    roof = {"kernel": _native.last_gather_kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": int(alg_bytes), "algorithmic_bytes_note": f"per node {H * 4} (P row) + {H * 4} (out row)"
            + (f" + {arg_bytes} (arg, uint8)" if arg_bytes else "") + f" + ids ({id_note})",
            "avg_launch_us": round(ms * 1e3, 2), "event_bracket_overhead_us": round(ev_overhead_ms * 1e3, 2),
            "launches": ksum[gname][0],
            "cache_state": "inside the training step: P/Q written by the preceding launch (L2 / Infinity-Cache resident), "
                           "ids and outputs cold; FETCH_SIZE counts Infinity-Cache hits, so `frac` is not a pure DRAM figure"}
    # HBM-side traffic from PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, profiles/README.md)
    # cannot be collected from inside this process: quoted from the committed summary of this round, if present
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_gather_max.json")))
    if args.dtype == "f32" and args.graph == "dynamic" and args.ragged is None and tpaths:
        try:
            roof["traffic"] = json.load(open(tpaths[-1])).get("hbm_bytes_per_launch")
            roof["traffic_source"] = (f"profiles/{os.path.basename(tpaths[-1])} (rocprofv3 PMC passes of this bench command, "
                                      "collected by tools/collect_profiles.sh: a committed figure, not measured by this process)")
        except Exception:
            pass
    # untimed legs: the same kernel form standalone, warm and cold
    if gname == "gather_max" and args.dtype == "f32":
        try:
            with torch.no_grad():
                from deepmetv2_amd import conv as conv_mod
                lds = conv_mod._lds_eligible(emb, lin.weight, table) if table.cnt is None else (table.max_nodes or 1 << 30) <= conv_mod._LDS_MAX_EVENT_NODES
                P, Q = _native.node_linear_split(emb, lin.weight, lin.bias, sliced=lds)
                want_arg = args.mode == "train"

                step_kernel = _native.last_gather_kernel
                rows16 = getattr(table, "rows16", None)
                winner_ids = table.cnt is not None and lds and want_arg and table.nonempty and rows16 is not None

                def launch():
                    if winner_ids:      # the form the static flow's training step runs (conv._EdgeConvLinearMax)
                        return _native.gather_max_local_j16(P, Q, rows16, table.cnt, table.order_by_count(), table.ptr,
                                                            table.k, lds)
                    return _native.gather_max(P, Q, table.nbr, table.ptr, want_arg=want_arg, cnt=table.cnt, lds=lds,
                                              nbr_local=table.nbr_local, sliced=lds)

                def timed(flush):
                    ts = []
                    junk = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev) if flush else None
                    for _ in range(12):
                        if flush:
                            junk.fill_(1.0)
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(); launch(); b.record()
                        torch.cuda.synchronize(dev)
                        ts.append(a.elapsed_time(b))
                    ts.sort()
                    return ts[len(ts) // 2]

                launch(); torch.cuda.synchronize(dev)
                roof["legs_kernel_matches_step"] = _native.last_gather_kernel == step_kernel
                warm_ms, cold_ms = timed(False), timed(True)
                roof["standalone_warm"] = {"us": round(warm_ms * 1e3, 2), "frac": round(alg_bytes / (warm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                roof["cold_us"] = round(cold_ms * 1e3, 2)
                roof["cold_frac"] = round(alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                roof["cold_note"] = "median of 12 launches, each after a 512 MB fill (L2 and Infinity Cache flushed); HIP events, bracket cost not removed"
        except Exception as e:  # the legs are diagnostics: never lose the bench line over them
            roof["cold_frac"] = None
            roof["cold_note"] = f"not measured: {type(e).__name__}: {e}"
    return roof


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU path in the product)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    # one process per GPU.  Under torch.distributed.run the group is created for ANY world size, also 1, so that the
    # N = 1 point of a scaling run executes the same RCCL calls (broadcast, all_reduce, barrier) as the N = 8 point
    use_group = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_group:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _StdoutToStderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(args.backend)
            dist.barrier()          # the communicator exists (and has said what it has to say) before anything is timed

    import deepmetv2_amd as dm
    from deepmetv2_amd import _native, synth
    from deepmetv2_amd.model import Net, loss_fn, split_features
    from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

    stock = args.model != "fused"
    if stock and (world > 1 or args.mode != "train" or args.input != "device" or args.dtype != "f32"
                  or args.graph == "static-table"):
        raise SystemExit("--model stock-* times the reference's own single-process training loop: N = 1, train, fp32, "
                         "--graph dynamic | static")
    # --hip-graph unset: "auto" -- on for N > 1; at N = 1 decided after the run-in from the measured host / GPU ratio (the
    # eager step costs the launch thread 0.9-1.3 ms box to box against 1.46 ms of GPU time: on a slow host the eager loop is
    # host-bound and the same kernels read 7 % slower)
    graph_capable = bool(args.mode == "train" and args.input == "device" and args.graph != "static" and not stock)
    hip_graph_auto = args.hip_graph is None and graph_capable
    if args.hip_graph is None:
        args.hip_graph = bool(world > 1 and graph_capable)
    hip_graph_note = None
    B, n, k = args.events_per_gpu, args.nodes, args.k
    sizes = [n] * B if args.ragged is None else synth.ragged_sizes(B, args.ragged[0], args.ragged[1], seed=1234 + rank)
    x, y, batch, ptr = synth.make_events(sizes, seed=1234 + rank, device=dev)
    dm.register_batch(batch, ptr, B, max_nodes=max(sizes), min_nodes=min(sizes))
    N = x.shape[0]

    torch.manual_seed(0)
    if stock:
        from deepmetv2_amd import stock_model
        variant = "static" if args.graph == "static" else args.model[len("stock-"):].replace("-", "_")
        model = stock_model.StockNet(dm, 8, 3, variant=variant, k=k).to(dev)
        if args.accelerate == "layers":
            model = dm.accelerate(model, fuse=False)
        elif args.accelerate == "fuse":
            model = dm.accelerate(model, graph="static" if variant == "static" else "dynamic", k=k)
        flat = sync = None
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)          # train.py:75 as written
        optimizer_name = "torch.optim.AdamW(model.parameters(), lr=1e-3)"
    else:
        model = Net(8, 3, graph="dynamic" if args.graph == "dynamic" else "static", k=k, edge_dtype=torch.bfloat16 if args.dtype == "bf16" else None).to(dev)
        flat = FlatModule(model)
        sync = GradSync(flat)
        sync.broadcast_state(0)
        # AdamW (train.py:75) on the flat parameter vector: one launch (deepmetv2_amd.optim.FlatAdamW = dmet_adamw_lr_f32,
        # device-side step counter and learning rate: capturable); --optimizer torch: torch.optim.AdamW(fused=True), two
        # launches, 16 us more per step
        if args.optimizer == "torch":
            opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True, capturable=args.hip_graph)
            optimizer_name = "torch.optim.AdamW(fused=True) on the flat parameter tensor"
        else:
            from deepmetv2_amd.optim import FlatAdamW
            opt = FlatAdamW([flat.flat_param], lr=1e-3)
            optimizer_name = "deepmetv2_amd.optim.FlatAdamW (one HIP launch) on the flat parameter tensor"

    if args.mode == "train":
        model.train()

        def static_graph():
            if args.graph == "dynamic":
                return None
            phi = torch.atan2(x[:, 1], x[:, 0])                                   # train.py:45-48
            etaphi = torch.cat([x[:, 3][:, None], phi[:, None]], dim=1)
            if args.graph == "static-table":
                return dm.radius_table(etaphi, r=0.4, batch=batch, loop=True, max_num_neighbors=255)
            return dm.radius_graph(etaphi, r=0.4, batch=batch, loop=True, max_num_neighbors=255)

        static_graph_sync = static_graph
        if args.graph == "static-table" and args.graph_async:
            # inside a CAPTURED step the table is built on a side stream beside the model's encoder (which does not need
            # it) and joined by the first EdgeConv: fork / join are graph dependencies there (0.857 -> 0.848 ms/step).  In
            # the eager loop the same fork / join are cross-queue barriers and the two queues time-slice: 0.94 -> 1.24 ms
            def static_graph():
                return dm.build_async(static_graph_sync) if args.hip_graph else static_graph_sync()

        if stock:
            def step():
                return stock_model.stock_train_step(dm, model, opt, x, y, batch,
                                                    graph_fn=(lambda _x: static_graph()) if args.graph == "static" else None)
        elif args.input == "host":
            if args.hip_graph or args.graph != "dynamic":
                raise SystemExit("--input host is built for the eager dynamic flow")
            import itertools
            from deepmetv2_amd.data import Batch, DeviceLoader
            hb = Batch(x.cpu(), y.cpu(), batch.cpu(), ptr.cpu(), max(sizes), min_nodes=min(sizes)).pin_memory()
            feed = iter(DeviceLoader(itertools.repeat(hb), dev, depth=2))     # the same batch every step: rate only

            def step():
                b = next(feed)
                return train_step(model, flat, sync, opt, b.x, b.y, b.batch, b.ptr)
        else:
            if args.hip_graph and args.graph == "static":
                raise SystemExit("--hip-graph: radius_graph sizes its [2,E] result on the host (one sync per step); "
                                 "use --graph static-table")

            def eager_train_step():
                return train_step(model, flat, sync, opt, x, y, batch, ptr, edge_index=static_graph())

            def graphed_train_step():
                """The step as two hipGraphs around the collective, or None (with a note) when the capture fails: a
                scaling point is never lost over it."""
                nonlocal hip_graph_note
                from deepmetv2_amd.parallel import GraphedTrainStep
                try:
                    return GraphedTrainStep(model, flat, sync, opt, x, y, batch, ptr,
                                            graph_fn=(lambda _x: static_graph()) if args.graph == "static-table" else None)
                except Exception as e:
                    hip_graph_note = f"capture failed ({type(e).__name__}: {e}); eager step timed instead"
                    return None

            step = eager_train_step
            if args.hip_graph:
                step = graphed_train_step() or eager_train_step
                args.hip_graph = step is not eager_train_step
    else:
        model.eval()

        def step():
            with torch.no_grad():
                xc, xk = split_features(x)
                w = model(xc, xk, None, batch)
                return dm.met_reduce(w, x, ptr=ptr)

    def barrier():
        if use_group:
            dist.barrier()
        torch.cuda.synchronize(dev)

    prewarm_steps = 0
    if args.prewarm_ms > 0:
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            prewarm_steps += 8
    if hip_graph_auto and not args.hip_graph:
        # host-bound?  time the launch thread's enqueue of one eager step from an idle stream against the GPU's time per
        # step over a short burst; above 0.5 the timed region replays hipGraphs instead (all ranks decide alike)
        hts = []
        for _ in range(7):
            torch.cuda.synchronize(dev)
            th = time.perf_counter(); step(); hts.append(time.perf_counter() - th)
        torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        for _ in range(24):
            step()
        torch.cuda.synchronize(dev)
        ratio = sorted(hts)[3] / ((time.perf_counter() - tg) / 24)
        # 0.5, not the 0.8 of the first session: at 0.8 the launch thread has no headroom left -- one box measured 0.80 in
        # this probe, stayed eager and then ran the 300 timed steps at 1.675 ms (38.2k events/s) where the replayed step
        # takes 1.38 (46.3k): any stall of the launch thread (allocator, Python) is GPU idle time it never catches up on
        want = ratio > 0.5
        if use_group:
            flag = torch.tensor([1.0 if want else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            want = bool(flag.item() > 0)
        hip_graph_note = f"auto: host enqueue / step time of the eager loop = {ratio:.2f} ({'>' if want else '<='} 0.5)"
        if want:
            g = graphed_train_step()
            if g is not None:
                step, args.hip_graph = g, True
                for _ in range(8):
                    step()
                torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    ev_overhead_ms = _native.timer.calibrate(dev)
    # inside the timed region only the graded kernel is bracketed (a bracket is two extra stream commands, ~5-10 us
    # of GPU idle each); the other native calls are bracketed in a few extra, untimed steps afterwards
    _native.timer.only = {"gather_max", "edgeconv_fused"}
    _native.timer.enabled = True
    _native.timer.reset()
    import gc
    # no collection inside the timed loop (it stalls the launch thread for a step or more) -- and none right before it
    # either: after a gc.collect() the first step takes the launch thread 2.2 ms instead of 1.2 and the GPU waits
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    dbg = [] if os.environ.get("DMET_BENCH_STEP_TIMES") == "1" else None
    for it in range(args.steps):
        _native.timer.enabled = it % args.bracket_every == 0
        step()
        if dbg is not None:
            dbg.append(time.perf_counter() - t0)
    barrier()
    elapsed = time.perf_counter() - t0
    if dbg is not None:
        print("host enqueue time per step (ms):", [round(1e3 * (b - a), 3) for a, b in zip([0.0] + dbg[:-1], dbg)][:25],
              "total", round(1e3 * elapsed, 3), file=sys.stderr)
    gc.enable()
    _native.timer.enabled = False
    ksum = _native.timer.summary()
    roof_leg = None
    # host side: the launch thread's time to ENQUEUE one step with nothing queued in front of it (synchronise, then time
    # one call).  Eager: ~55 launches through Python; --hip-graph: two graph replays + the collective.  When this is
    # close to ms_per_step the run is host-bound and N ranks on one host will not scale.
    host_ts = []
    for _ in range(10):
        torch.cuda.synchronize(dev)
        th = time.perf_counter()
        step()
        host_ts.append(time.perf_counter() - th)
    torch.cuda.synchronize(dev)
    host_ts.sort()
    host_ms = host_ts[len(host_ts) // 2] * 1e3
    if args.hip_graph and args.mode == "train":
        # kernels inside a replayed hipGraph cannot be bracketed with events: the per-kernel figures of a --hip-graph run
        # come from a few eager steps of the same training step after the timed region (said so in the roofline block)
        def eager_step():
            return train_step(model, flat, sync, opt, x, y, batch, ptr, edge_index=static_graph())
        eager_step(); torch.cuda.synchronize(dev)
        _native.timer.only = None
        _native.timer.enabled = True
        _native.timer.reset()
        for _ in range(min(5, args.steps)):
            eager_step()
        torch.cuda.synchronize(dev)
        _native.timer.enabled = False
        ksum = _native.timer.summary()
        roof_leg = "eager steps after the timed region (the timed steps replay hipGraphs, whose kernels cannot be bracketed)"
    if not args.hip_graph:
        _native.timer.only = None
        _native.timer.enabled = True
        _native.timer.reset()
        for _ in range(min(5, args.steps)):
            step()
        torch.cuda.synchronize(dev)
        _native.timer.enabled = False
        for name, v in _native.timer.summary().items():
            ksum.setdefault(name, v)
    if use_group:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        events = B * world * args.steps
        roof = gather_roofline(args, ksum, ev_overhead_ms, model, x, batch, ptr, sizes, static_graph if args.mode == "train" else None, dev)
        if roof is not None and stock and args.graph == "static":
            roof.pop("standalone_warm", None)
        if roof is not None and roof_leg is not None:
            roof["measured_in"] = roof_leg
        kernels = {}
        for name, (cnt, ms) in sorted(ksum.items()):
            kernels[name] = {"launches": cnt, "avg_us": round(ms * 1e3, 2)}
        if "knn" in ksum:
            # the whole graph build (prep + plans, matrix-core filter with in-place exact re-rank, tail merge, fallback
            # kernels) per call, against the stated floor of its design
            pairs = float(sum(sz * sz for sz in sizes))
            fl = knn_floor(sizes)
            kernels["knn"].update({"path": os.environ.get("DMET_KNN_PATH", "mfma_filter+exact_rerank"),
                                   "pairs_per_s": round(pairs / (ksum["knn"][1] * 1e-3), 1), **fl,
                                   "floor_us": max(fl["mfma_floor_us"], fl["valu_floor_us"]),
                                   "floor_note": "max(mfma, valu): the two pipes overlap once the matrix share is small "
                                                 "(profiles/r02_knn_filter2_budget.md section 4)",
                                   "frac_of_floor": round(max(fl["mfma_floor_us"], fl["valu_floor_us"]) / (ksum["knn"][1] * 1e3), 3)})
            from deepmetv2_amd import conv as _conv
            riders = []
            if _conv.KNN_RIDER != "0" and args.graph == "dynamic":
                riders.append("the EdgeConv's node-level dense layer (trailing workgroups of the filter launch; no node_linear_split launch)")
            if _conv.BN_KNN_FUSE != "0" and args.graph == "dynamic" and args.mode == "train":
                riders.append("the BatchNorm transform + residual add that produces the build's input (inside the prep launch; no bn_apply launch)")
            if riders:
                kernels["knn"]["also_carries"] = riders   # the bracket times that work too: the floor fraction is a lower bound
        out = {
            "metric": "events/sec (4.5k PF cands, k=16)", "value": round(events / elapsed, 1), "unit": "events/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "host_ms_per_step": round(host_ms, 3),
            "host_over_gpu": round(host_ms / (elapsed / args.steps * 1e3), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_label(args, B, n, k), "events_per_gpu": B,
                       "nodes_per_event": n if args.ragged is None else f"U[{args.ragged[0]},{args.ragged[1]}]", "k": k,
                       "global_batch": B * world, "mode": args.mode, "graph": args.graph, "parallelism": f"dp{world}",
                       "hip_graph": bool(args.hip_graph), "input": args.input,
                       "model": args.model + ("" if args.accelerate == "none" else f" + accelerate({args.accelerate})"),
                       "optimizer": optimizer_name, "prewarm_ms": args.prewarm_ms, "gc_disabled": True,
                       "graded_kernel_bracketed_every_nth_step": args.bracket_every,
                       "weight_grad_sums": "one launch per step (dmet_finalize_flush)" if _native.DEFER_FINALIZE else "per call"},
            "roofline": roof, "kernels": kernels,
        }
        if hip_graph_note:
            out["config"]["hip_graph_note"] = hip_graph_note
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, seed=1234)
            if args.graph == "dynamic" and args.dtype == "f32" and not stock:
                out["parity"] = parity_sample(args, model, dev, seed=4321)
        print(json.dumps(out), flush=True)
    if use_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
