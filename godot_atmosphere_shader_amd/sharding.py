"""Sharding of framebuffer work across the GPUs of one node, and the gather of the shaded pixels.

Every pixel is independent and all inputs are read-only (uniforms + ~1.1 MB of textures replicated per
GPU), so there is no exchange during compute.  Two partitions are supported:

  * viewports: N independent viewports, one per rank (BASELINE configs[4]; weak scaling);
  * row bands: one viewport cut into contiguous bands of rows, one per rank, so that each rank's output
    is one contiguous slab of the frame (strong scaling).  `balanced_row_bands` cuts by per-row hit
    counts instead of row counts when bands would otherwise be unequal work (P_space has empty rows).

The only collective is the final gather of RGBA32F pixels to the root rank (`torch.distributed`, backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Root ingress is per-link bound
(7 xGMI links x ~153 GB/s), a direct gather -- not a ring -- is the right shape.  `FrameGather` keeps
`depth` gathers in flight so the gather of frame k overlaps the render of frame k+1.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def row_bands(height: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [y0, y1) bands, as equal as possible; empty bands when world_size > height."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    return [(height * k // world_size, height * (k + 1) // world_size) for k in range(world_size)]


def balanced_row_bands(row_cost: Sequence[float], world_size: int) -> List[Tuple[int, int]]:
    """Cut rows into `world_size` contiguous bands of (nearly) equal total cost (e.g. hit pixels per row).
    Falls back to equal row counts when the total cost is zero."""
    cost = np.asarray(row_cost, dtype=np.float64)
    h = len(cost)
    total = float(cost.sum())
    if total <= 0.0:
        return row_bands(h, world_size)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for k in range(1, world_size):
        target = total * k / world_size
        y = int(np.searchsorted(cum, target, side="left"))
        cuts.append(min(max(y, cuts[-1]), h))
    cuts.append(h)
    return [(cuts[k], cuts[k + 1]) for k in range(world_size)]


def band_rect(width: int, band: Tuple[int, int]) -> Tuple[int, int, int, int]:
    return (0, band[0], width, band[1])


class FrameGather:
    """Gathers per-rank pixel slabs to rank `dst`.

    viewports mode: every rank contributes a full (H, W, 4) frame; root receives (world, H, W, 4).
    bands mode: rank r contributes rows bands[r] of one frame; root receives them in place in (H, W, 4)
    (row bands are contiguous slabs of the frame, so the receive buffers are views -- no extra copy).
    Unequal bands are padded to the tallest band for the collective and trimmed on the root.
    """

    def __init__(self, height: int, width: int, device, dst: int = 0, bands=None, depth: int = 2, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.dst = dst
        self.h, self.w = height, width
        self.bands = list(bands) if bands is not None else None
        self.depth = max(1, depth)
        self.device = device
        self._inflight = []  # (work, slot)
        self._slot = 0
        if self.bands is None:
            self.send_shape = (height, width, 4)
        else:
            if len(self.bands) != self.world:
                raise ValueError("need one band per rank")
            self.max_rows = max(b[1] - b[0] for b in self.bands)
            self.equal = all((b[1] - b[0]) == self.max_rows for b in self.bands)
            self.send_shape = (self.max_rows, width, 4)
        self.send = [torch.empty(self.send_shape, dtype=torch.float32, device=device) for _ in range(self.depth)]
        self.recv = None
        if self.rank == dst:
            if self.bands is None:
                self.recv = [torch.empty((self.world,) + self.send_shape, dtype=torch.float32, device=device)
                             for _ in range(self.depth)]
            elif self.equal:
                self.recv = [torch.empty((height, width, 4), dtype=torch.float32, device=device) for _ in range(self.depth)]
            else:
                self.recv = [torch.empty((self.world,) + self.send_shape, dtype=torch.float32, device=device)
                             for _ in range(self.depth)]
                self.frames = [torch.empty((height, width, 4), dtype=torch.float32, device=device) for _ in range(self.depth)]

    def _root_slot(self, slot: int):
        """On the root: the part of the receive buffer its own contribution lands in -- rendering straight into it makes the
        root's share of the gather a no-op (dist.gather copies tensor -> gather_list[dst] on the root only when they differ;
        a root that rendered into a separate send buffer paid one device-to-device copy of its whole frame per gather:
        15.4 against 22.2 Grays/s on a 1-rank group in round 2).  None where the layouts differ (unequal bands are padded)."""
        if self.rank != self.dst:
            return None
        r = self.recv[slot]
        if self.bands is None:
            return r[self.rank]
        if self.equal:
            b = self.bands[self.rank]
            return r[b[0]:b[1]]
        return None

    def next_send_buffer(self):
        """Buffer to render the next contribution into.  Waits (stream-side) for the gather that last used it."""
        slot = self._slot
        while len(self._inflight) >= self.depth:
            work, _ = self._inflight.pop(0)
            work.wait()
        own = self._root_slot(slot)
        if own is not None:
            return own, slot
        buf = self.send[slot]
        if self.bands is not None:
            rows = self.bands[self.rank][1] - self.bands[self.rank][0]
            return buf[:rows], slot
        return buf, slot

    def submit(self, slot: int):
        """Start the gather of send buffer `slot` (asynchronously; ordered after work already enqueued on the
        current stream)."""
        dist = self.dist
        gather_list = None
        tensor = self.send[slot]
        if self.rank == self.dst:
            r = self.recv[slot]
            if self.bands is not None and self.equal:
                gather_list = [r[b[0]:b[1]] for b in self.bands]
            else:
                gather_list = [r[k] for k in range(self.world)]
            own = self._root_slot(slot)
            if own is not None:
                tensor = own                   # rendered in place ...
                gather_list[self.rank] = own   # ... and the very same tensor object: the root's self-copy is skipped
        work = dist.gather(tensor, gather_list, dst=self.dst, group=self.group, async_op=True)
        self._inflight.append((work, slot))
        self._slot = (slot + 1) % self.depth
        return work

    def finish(self):
        """Wait for every gather in flight; on the root returns the most recent result
        ((world, H, W, 4) in viewports mode, (H, W, 4) in bands mode), elsewhere None."""
        last = None
        for work, slot in self._inflight:
            work.wait()
            last = slot
        self._inflight = []
        if self.rank != self.dst or last is None:
            return None
        if self.bands is None or self.equal:
            return self.recv[last]
        frame = self.frames[last]
        for k, (y0, y1) in enumerate(self.bands):
            frame[y0:y1] = self.recv[last][k, : y1 - y0]
        return frame
