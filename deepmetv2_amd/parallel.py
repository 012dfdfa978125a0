"""Data parallelism over events: one process per GPU, a single flat gradient all-reduce per step (RCCL over xGMI).

The reference is single-process (train.py:72); events are independent graphs (neighbourhoods never cross the
`batch` boundaries, MET is per event), so the batch shards with no exchange in forward.  The whole model has 6 641
parameters: the per-step collective is ONE all-reduce of a 26.6 KB fp32 buffer, latency-bound, so there is no
bucketing or overlap machinery -- parameters and gradients live in two flat buffers and the collective runs on
the flat gradient right after backward.  BatchNorm statistics stay per rank.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(num_items: int, rank: int, world: int) -> range:
    """Contiguous block of items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def balanced_shards(costs: Sequence[float], world: int) -> List[List[int]]:
    """Ragged events: greedy longest-first assignment by cost (kNN cost ~ n_b^2), deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda w: (loads[w], w))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in out]


class FlatModule:
    """Re-homes every parameter (and its gradient) of `module` into two contiguous fp32 buffers.

    `flat_param` aliases the module's parameters and `flat_grad` collects their gradients after every backward
    (`gather_grads`), so one optimizer tensor and one all_reduce() cover the whole model.  Parameter names/shapes
    (state_dict) are unchanged.
    """

    def __init__(self, module: torch.nn.Module):
        self.module = module
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        # every parameter starts on a 16-byte boundary (4 floats): the fused kernels take 16-byte loads of weights and
        # BatchNorm vectors and fall back to slower forms otherwise.  The padding elements are zero, get zero gradients
        # and stay zero under AdamW; the reference model's sizes are all multiples of 4 except the last bias, so its
        # flat buffer has no padding at all (6 641 elements)
        offsets, total = [], 0
        for p in params:
            total = (total + 3) // 4 * 4
            offsets.append(total)
            total += p.numel()
        self.flat_param = torch.nn.Parameter(torch.zeros(total, device=dev, dtype=dt))
        self.flat_param.grad = torch.zeros(total, device=dev, dtype=dt)
        self.grad_views = []
        with torch.no_grad():
            for p, off in zip(params, offsets):
                n = p.numel()
                self.flat_param.data[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param.data[off:off + n].view_as(p)
                self.grad_views.append(self.flat_param.grad[off:off + n].view_as(p))
                p.grad = None
        self.params = params
        self.offsets = offsets
        self.numel = total

    @property
    def flat_grad(self) -> torch.Tensor:
        return self.flat_param.grad

    def zero_grad(self) -> None:
        """Drop the per-parameter gradients: backward then hands each parameter a fresh tensor (no accumulate kernel
        per parameter), and `gather_grads` moves them into the flat buffer in one multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def gather_grads(self) -> None:
        """flat_grad <- the parameters' gradients of the last backward (zeros where a parameter got none).  Weight-gradient
        sums that the backward pass left queued (`_native.finalize_defer_begin`) are formed first, in one launch."""
        from . import _native
        _native.finalize_flush()
        pairs = [(v, p.grad) for v, p in zip(self.grad_views, self.params) if p.grad is not None]
        if len(pairs) != len(self.params):
            self.flat_param.grad.zero_()
        if pairs:
            torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])

    def buffers(self) -> List[torch.Tensor]:
        return [b for b in self.module.buffers()]


class GradSync:
    """DDP semantics for a FlatModule: broadcast parameters+buffers from rank 0 once, then combine the gradients.

    The loss is a MEAN over events (model/net.py:60), so the gradient of the global batch is
    sum_r (B_r / B) grad_r with B_r the events of rank r and grad_r the gradient of that rank's own mean.  Each rank
    therefore seeds its backward pass with its share B_r / B (`loss_seed`; 1 / world unless `set_share` says otherwise:
    shards balanced by cost hold unequal event counts) and the collective is a plain SUM -- no division kernel after
    the all-reduce, and unequal shards are weighted correctly."""

    def __init__(self, flat: FlatModule, group: Optional[dist.ProcessGroup] = None):
        self.flat = flat
        self.group = group
        # a process group of ONE rank still runs its collectives (so that `torchrun --nproc-per-node 1` exercises the
        # same RCCL calls as N ranks do); without any group the step has no collective at all
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self._share = 1.0 / self.world
        self._seed: Optional[torch.Tensor] = None

    def set_share(self, local_events: int, global_events: int) -> None:
        """This rank holds `local_events` of the `global_events` events of the step's global batch."""
        if local_events < 0 or global_events <= 0 or local_events > global_events:
            raise ValueError(f"bad shares: {local_events} of {global_events} events")
        share = local_events / global_events
        if share != self._share:
            self._share = share
            if self._seed is not None:
                self._seed.fill_(share)      # in place: a captured backward pass reads this tensor

    def loss_seed(self, loss: torch.Tensor) -> torch.Tensor:
        """d(global loss) / d(this rank's loss): the gradient `loss.backward` starts from (a cached device scalar)."""
        if self._seed is None or self._seed.device != loss.device or self._seed.dtype != loss.dtype:
            self._seed = torch.full((), self._share, device=loss.device, dtype=loss.dtype)
        return self._seed

    def broadcast_state(self, src: int = 0) -> None:
        if not self.active:
            return
        dist.broadcast(self.flat.flat_param.data, src=src, group=self.group)
        for b in self.flat.buffers():
            dist.broadcast(b, src=src, group=self.group)

    def average_gradients(self) -> None:
        """One all-reduce(SUM) of the flat fp32 gradient (26.6 KB).  The ranks' backward passes were seeded with their
        shares of the global batch (`loss_seed`), so the sum IS the gradient of the global mean."""
        if not self.active:
            return
        dist.all_reduce(self.flat.flat_grad, op=dist.ReduceOp.SUM, group=self.group)


def train_step(model: torch.nn.Module, flat: FlatModule, sync: GradSync, optimizer: torch.optim.Optimizer,
               x: torch.Tensor, y: torch.Tensor, batch: torch.Tensor, ptr: Optional[torch.Tensor] = None,
               edge_index: Optional[torch.Tensor] = None, global_events: Optional[int] = None) -> torch.Tensor:
    """One training step, the sequence of /root/reference/train.py:40-52:
    zero_grad -> split features -> model -> loss_fn -> backward -> (all-reduce) -> optimizer.step.
    `global_events`: events of the step's global batch over all ranks when the ranks hold unequal numbers of them
    (`data.EventLoader(..., rank, world, balance="cost")` reports it as `Batch.global_graphs`); None: equal shares."""
    from .model import loss_fn, split_features

    if global_events is not None:
        sync.set_share(int(y.shape[0]), int(global_events))
    flat.zero_grad()
    x_cont, x_cat = split_features(x, lazy_cat=True)
    weights = model(x_cont, x_cat, edge_index, batch)
    loss = loss_fn(weights, x, y, batch, ptr=ptr)
    from . import _native
    _native.finalize_defer_begin()       # the weight-gradient sums of the whole pass: one launch inside gather_grads
    try:
        loss.backward(sync.loss_seed(loss))
    finally:
        flat.gather_grads()
    sync.average_gradients()
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """The training step as two hipGraphs (torch.cuda.CUDAGraph) around the one collective:
    graph A = zero_grad .. backward .. gather_grads, then the (eager) gradient all-reduce, then graph B = optimizer
    step.  Pays off when the step is launch-bound (small batches: ~100 launches per step); at BASELINE config 2 the
    step is GPU-bound and the two forms time the same.  The inputs are static buffers: refill them with `load`.
    The optimizer must be built with capturable=True; batch metadata must be registered (no host sync in forward)."""

    def __init__(self, model, flat: FlatModule, sync: GradSync, optimizer, x, y, batch, ptr=None, warmup: int = 3,
                 graph_fn=None):
        """graph_fn(x) -> the static graph of the batch (the reference's active flow, train.py:45-48), built INSIDE the
        captured step; it must not synchronise with the host, i.e. return a NeighborTable (dm.radius_table), not
        radius_graph's [2,E] tensor.  None: the model builds its kNN graphs itself (dynamic flow)."""
        from .model import loss_fn, split_features
        self.flat, self.sync, self.opt = flat, sync, optimizer
        self.x, self.y, self.batch, self.ptr = x, y, batch, ptr

        def fwd_bwd():
            flat.zero_grad()
            x_cont, x_cat = split_features(self.x, lazy_cat=True)
            graph = graph_fn(self.x) if graph_fn is not None else None
            loss = loss_fn(model(x_cont, x_cat, graph, self.batch), self.x, self.y, self.batch, ptr=self.ptr)
            from . import _native
            _native.finalize_defer_begin()
            try:
                loss.backward(sync.loss_seed(loss))
            finally:
                flat.gather_grads()
            return loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd()
                sync.average_gradients()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        # With a process group alive, its watchdog THREAD polls the events of outstanding collectives (here: the warm-up's
        # all-reduces).  Under the default capture mode ("global") an event query from any thread while this one captures
        # is an error (hipErrorStreamCaptureUnsupported) that the watchdog turns into process termination -- seen once
        # in ~25 runs of tests/test_gpu_rccl.py.  So: nothing of ours outstanding when the capture starts, and a capture
        # mode that only polices the capturing thread.
        mode = "global"
        if sync.active:
            torch.cuda.synchronize()
            mode = "thread_local"
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, capture_error_mode=mode):
            self.loss = fwd_bwd()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, capture_error_mode=mode):
            optimizer.step()

    def load(self, x, y) -> None:
        """Copy a new batch of the same shape into the captured input buffers."""
        self.x.copy_(x); self.y.copy_(y)

    def __call__(self) -> torch.Tensor:
        self.graph_a.replay()
        self.sync.average_gradients()
        # a scheduler may have changed the learning rate since the last replay (train.py:58,76): optimizers that keep
        # it on the device (optim.FlatAdamW) push the new value before the captured step runs
        push = getattr(self.opt, "sync_hyper", None)
        if push is not None:
            push()
        self.graph_b.replay()
        return self.loss
