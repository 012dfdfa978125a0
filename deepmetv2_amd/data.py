"""Ragged event batches: what PyG's Data / Batch / DataLoader collate give the reference.

The reference stores one `Data(x[n,11], y[1,Y])` per event (/root/reference/model/data_loader.py:63-90) and lets the
PyG DataLoader concatenate them into `Batch(x[N,11], y[B,Y], batch[N])` (:107-110).  Here the same container also
carries `ptr[B+1]` and the largest event size, and registers them with the operators when it is moved to the GPU,
so no kernel launch ever needs a device->host sync to learn the batch structure.

Wire format of the raw files (data_*/generate_npz.py:125-140): `x[12, n_evt, n_max]` padded with -999 in the order
pt, eta, phi, d0, dz, mass, puppiWeight, pdgId, charge, fromPV, pvRef, pvAssocQuality; `y[n_evt, Y]`.
"""
from __future__ import annotations

from dataclasses import dataclass
import collections
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graph import register_batch

PAD = -999.0
CLIP = 5000.0
# columns of the model input (data_loader.py:72): pX, pY, pT, eta, d0, dz, mass, puppiWeight, pdgId, charge, fromPV
FEATURES = ("pX", "pY", "pT", "eta", "d0", "dz", "mass", "puppiWeight", "pdgId", "charge", "fromPV")


@dataclass
class Batch:
    x: torch.Tensor        # [N, 11] float32
    y: torch.Tensor        # [B, Y]  float32
    batch: torch.Tensor    # [N] int64, sorted
    ptr: torch.Tensor      # [B+1] int64
    max_nodes: int
    # events of the GLOBAL batch this one is a rank's shard of (None: not sharded); parallel.train_step weights the
    # rank's gradient by num_graphs / global_graphs
    global_graphs: Optional[int] = None
    min_nodes: Optional[int] = None     # smallest event (>= k: kNN tables of this batch have no empty slot to look for)

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel() - 1)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def to(self, device, non_blocking: bool = False) -> "Batch":
        b = Batch(self.x.to(device, non_blocking=non_blocking), self.y.to(device, non_blocking=non_blocking),
                  self.batch.to(device, non_blocking=non_blocking), self.ptr.to(device, non_blocking=non_blocking),
                  self.max_nodes, self.global_graphs, self.min_nodes)
        register_batch(b.batch, b.ptr, b.num_graphs, max_nodes=b.max_nodes, min_nodes=b.min_nodes)
        return b

    def pin_memory(self) -> "Batch":
        return Batch(self.x.pin_memory(), self.y.pin_memory(), self.batch.pin_memory(), self.ptr.pin_memory(),
                     self.max_nodes, self.global_graphs, self.min_nodes)


def collate(events: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> Batch:
    """[(x_e[n_e,11], y_e[1,Y] or [Y]), ...] -> Batch, nodes concatenated in event order (PyG Batch semantics)."""
    if len(events) == 0:
        raise ValueError("cannot collate an empty list of events")
    xs = [e[0] for e in events]
    ys = [e[1].reshape(1, -1) for e in events]
    counts = torch.tensor([int(x.shape[0]) for x in xs], dtype=torch.int64)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    batch = torch.repeat_interleave(torch.arange(len(xs), dtype=torch.int64), counts)
    x = torch.cat(xs, 0).to(torch.float32).contiguous()
    return Batch(x, torch.cat(ys, 0).to(torch.float32).contiguous(), batch, ptr, int(counts.max()),
                 min_nodes=int(counts.min()))


def events_from_padded(x_pad, y) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Decode one raw file: padded `x_pad[12, n_evt, n_max]`, `y[n_evt, Y]` -> per-event (x[n,11], y[1,Y]).

    Same transformation as METDataset.process (data_loader.py:67-90), done for the whole file at once:
    pX = pt*cos(phi), pY = pt*sin(phi), keep candidates whose pdgId and charge are not the -999 padding,
    nan_to_num, clip to +-5000."""
    xp = np.asarray(x_pad, dtype=np.float32)
    yy = np.asarray(y, dtype=np.float32)
    if xp.ndim != 3 or xp.shape[0] < 10:
        raise ValueError(f"x_pad must be [>=10, n_evt, n_max], got {xp.shape}")
    pt, eta, phi = xp[0], xp[1], xp[2]
    feats = np.stack([pt * np.cos(phi), pt * np.sin(phi), pt, eta, xp[3], xp[4], xp[5], xp[6], xp[7], xp[8], xp[9]],
                     axis=-1)                                   # [n_evt, n_max, 11]
    keep = (feats[..., 8] != PAD) & (feats[..., 9] != PAD)
    feats = np.clip(np.nan_to_num(feats), -CLIP, CLIP).astype(np.float32)
    out = []
    for e in range(feats.shape[0]):
        out.append((torch.from_numpy(feats[e][keep[e]].copy()), torch.from_numpy(yy[e:e + 1].copy())))
    return out


class EventLoader:
    """Minimal stand-in for `DataLoader(subset, batch_size, shuffle=False)` (data_loader.py:107-110): yields Batches
    of consecutive events; `split()` reproduces the reference's seeded random train/validation split (:95-104).

    Data parallel (the role a `DistributedSampler` would play next to data_loader.py:107-110; the reference itself is
    single-process): with `world > 1` every rank walks the SAME sequence of global batches of `batch_size` events and
    yields its own shard of each -- `balance="count"`: contiguous blocks whose sizes differ by at most one
    (`parallel.shard_range`); `balance="cost"`: greedy longest-first by n^2, the kNN build's cost, for ragged batches
    (`parallel.balanced_shards`; ranks then hold unequal event counts).  Every shard carries `global_graphs`, from which
    `parallel.train_step(..., global_events=b.global_graphs)` weights the rank's gradient, so N ranks walk the gradient
    trajectory of the single-process global batch (up to per-rank BatchNorm statistics).  A trailing global batch with
    fewer events than ranks is dropped (a rank without events has no step to run)."""

    def __init__(self, events: Sequence[Tuple[torch.Tensor, torch.Tensor]], batch_size: int,
                 indices: Optional[Sequence[int]] = None, device: Optional[torch.device] = None,
                 rank: int = 0, world: int = 1, balance: str = "count"):
        if balance not in ("count", "cost"):
            raise ValueError(f"balance must be 'count' or 'cost', got {balance!r}")
        if world < 1 or not 0 <= rank < world:
            raise ValueError(f"bad rank / world: {rank} / {world}")
        if world > int(batch_size):
            raise ValueError(f"a global batch of {batch_size} events cannot be sharded over {world} ranks")
        self.events = events
        self.batch_size = int(batch_size)
        self.indices = list(range(len(events))) if indices is None else list(indices)
        self.device = device
        self.rank, self.world, self.balance = int(rank), int(world), balance

    def _global_batches(self) -> List[List[int]]:
        out = [self.indices[s:s + self.batch_size] for s in range(0, len(self.indices), self.batch_size)]
        if self.world > 1 and out and len(out[-1]) < self.world:
            out.pop()
        return out

    def __len__(self) -> int:
        return len(self._global_batches())

    def shard(self, ids: Sequence[int]) -> List[int]:
        """The events of one global batch that this rank computes (in the batch's order)."""
        if self.world == 1:
            return list(ids)
        from .parallel import balanced_shards, shard_range
        if self.balance == "count":
            return [ids[i] for i in shard_range(len(ids), self.rank, self.world)]
        costs = [float(self.events[i][0].shape[0]) ** 2 for i in ids]
        return [ids[i] for i in balanced_shards(costs, self.world)[self.rank]]

    def __iter__(self) -> Iterator[Batch]:
        for ids in self._global_batches():
            mine = self.shard(ids)
            if not mine:
                # cost balancing can leave a rank empty only if the batch has fewer events than ranks (dropped above)
                raise RuntimeError("a rank received no events of a global batch")
            b = collate([self.events[i] for i in mine])
            if self.world > 1:
                b.global_graphs = len(ids)
            yield b.to(self.device) if self.device is not None else b

    @staticmethod
    def split(events, batch_size: int, validation_split: float = 0.2, seed: int = 42, device=None, rank: int = 0,
              world: int = 1, balance: str = "count"):
        n = len(events)
        n_val = int(np.floor(validation_split * n))
        g = torch.Generator().manual_seed(seed)
        perm = torch.randperm(n, generator=g).tolist()
        train, val = perm[: n - n_val], perm[n - n_val:]
        return {"train": EventLoader(events, batch_size, train, device, rank, world, balance),
                "test": EventLoader(events, batch_size, val, device, rank, world, balance)}


class DeviceLoader:
    """Feeds host Batches to the GPU the way `for data in dataloader: data.to(device)` does in the reference
    (train.py:39-41), with the copy taken off the critical path: every batch is staged in pinned host memory and copied
    host->device on a side stream `depth` batches ahead of the one the caller is computing on; the compute stream only
    waits for the copy's event.  The batch structure (ptr, event count, largest event) is registered with the operators
    from the host-side values, so the forward pass needs no device->host sync.

    `batches` is any iterable of host `Batch` objects (an `EventLoader` without a device)."""

    def __init__(self, batches: Iterable[Batch], device, depth: int = 2):
        self.batches = batches
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceLoader copies to a ROCm device; got " + str(device))
        self.depth = max(1, int(depth))

    def __len__(self) -> int:
        return len(self.batches)  # type: ignore[arg-type]

    def __iter__(self) -> Iterator[Batch]:
        dev = self.device
        copy_stream = torch.cuda.Stream(dev)
        inflight = collections.deque()

        def stage(hb: Batch):
            pinned = hb if hb.x.is_pinned() else hb.pin_memory()     # torch's caching host allocator reuses the blocks
            with torch.cuda.stream(copy_stream):
                db = Batch(pinned.x.to(dev, non_blocking=True), pinned.y.to(dev, non_blocking=True),
                           pinned.batch.to(dev, non_blocking=True), pinned.ptr.to(dev, non_blocking=True), hb.max_nodes,
                           hb.global_graphs, hb.min_nodes)
                done = torch.cuda.Event()
                done.record(copy_stream)
            inflight.append((pinned, db, done))

        it = iter(self.batches)
        for hb in it:
            stage(hb)
            if len(inflight) > self.depth:
                yield self._hand_over(inflight.popleft(), dev)
        while inflight:
            yield self._hand_over(inflight.popleft(), dev)

    @staticmethod
    def _hand_over(item, dev) -> Batch:
        _pinned, db, done = item
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(done)                       # device-side wait: the host does not block
        for t in (db.x, db.y, db.batch, db.ptr):
            t.record_stream(cur)                   # allocated on the copy stream, used on the compute stream
        register_batch(db.batch, db.ptr, db.num_graphs, max_nodes=db.max_nodes, min_nodes=db.min_nodes)
        return db
