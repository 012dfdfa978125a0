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

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel() - 1)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def to(self, device, non_blocking: bool = False) -> "Batch":
        b = Batch(self.x.to(device, non_blocking=non_blocking), self.y.to(device, non_blocking=non_blocking),
                  self.batch.to(device, non_blocking=non_blocking), self.ptr.to(device, non_blocking=non_blocking),
                  self.max_nodes)
        register_batch(b.batch, b.ptr, b.num_graphs, max_nodes=b.max_nodes)
        return b

    def pin_memory(self) -> "Batch":
        return Batch(self.x.pin_memory(), self.y.pin_memory(), self.batch.pin_memory(), self.ptr.pin_memory(),
                     self.max_nodes)


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
    return Batch(x, torch.cat(ys, 0).to(torch.float32).contiguous(), batch, ptr, int(counts.max()))


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
    of consecutive events; `split()` reproduces the reference's seeded random train/validation split (:95-104)."""

    def __init__(self, events: Sequence[Tuple[torch.Tensor, torch.Tensor]], batch_size: int,
                 indices: Optional[Sequence[int]] = None, device: Optional[torch.device] = None):
        self.events = events
        self.batch_size = int(batch_size)
        self.indices = list(range(len(events))) if indices is None else list(indices)
        self.device = device

    def __len__(self) -> int:
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Batch]:
        for s in range(0, len(self.indices), self.batch_size):
            b = collate([self.events[i] for i in self.indices[s:s + self.batch_size]])
            yield b.to(self.device) if self.device is not None else b

    @staticmethod
    def split(events, batch_size: int, validation_split: float = 0.2, seed: int = 42, device=None):
        n = len(events)
        n_val = int(np.floor(validation_split * n))
        g = torch.Generator().manual_seed(seed)
        perm = torch.randperm(n, generator=g).tolist()
        train, val = perm[: n - n_val], perm[n - n_val:]
        return {"train": EventLoader(events, batch_size, train, device),
                "test": EventLoader(events, batch_size, val, device)}


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
                           pinned.batch.to(dev, non_blocking=True), pinned.ptr.to(dev, non_blocking=True), hb.max_nodes)
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
        register_batch(db.batch, db.ptr, db.num_graphs, max_nodes=db.max_nodes)
        return db
