"""Sharding of framebuffer work across the GPUs of one node, and the gather of the shaded pixels.

Every pixel is independent and all inputs are read-only (uniforms + ~1.1 MB of textures replicated per
GPU), so there is no exchange during compute.  Two partitions are supported:

  * viewports: N independent viewports, one per rank (BASELINE configs[4]; weak scaling);
  * row bands: one viewport cut into contiguous bands of rows, one per rank, so that each rank's output
    is one contiguous slab of the frame (strong scaling).  `balanced_row_bands` cuts by per-row hit
    counts instead of row counts when bands would otherwise be unequal work (P_space has empty rows).
  * tile strips (round 4): one viewport cut into strips of STRIP_TILE_ROWS tile rows (16 pixel rows), dealt to the
    ranks longest-processing-time-first by their MEASURED cost (`lpt_strips`), so that every rank gets a mix of
    heavy and cheap tiles instead of one neighbourhood of the picture; a rank draws the tiles of its strips in ONE
    launch (atmo_render_tiles, heaviest tile first) and `StripGather` puts the strips back in place on the root.

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


# ---- tile strips (BASELINE north_star: "independent framebuffer tiles shard embarrassingly across the 8 GPUs") -------------------------
STRIP_TILE_ROWS = 2  # tile rows per strip: 2 x 8 = 16 pixel rows


def lpt_strips(tile_cost, world_size: int, strip_tile_rows: int = STRIP_TILE_ROWS):
    """Deals the strips of a tile grid to `world_size` ranks, longest processing time first.

    tile_cost: (tiles_y, tiles_x) measured costs (PlanetAtmosphere.measure_tile_costs).  A strip is `strip_tile_rows` consecutive
    tile rows; its cost the sum of its tiles'.  Strips are sorted by cost, heaviest first, and each goes to the rank with the least
    work so far (ties: the lowest rank) -- deterministic, so every rank computes the same deal from the same costs.
    Returns (strips_of_rank, tiles_of_rank): per rank the strip indices in ascending order, and the tile indices (row-major in the
    grid) of those strips sorted by tile cost, heaviest first (the order atmo_render_tiles draws them in)."""
    cost = np.asarray(tile_cost, dtype=np.float64)
    ty, tx = cost.shape
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    n_strips = (ty + strip_tile_rows - 1) // strip_tile_rows
    strip_cost = np.array([cost[k * strip_tile_rows:(k + 1) * strip_tile_rows].sum() for k in range(n_strips)])
    order = sorted(range(n_strips), key=lambda k: (-strip_cost[k], k))
    load = [0.0] * world_size
    strips = [[] for _ in range(world_size)]
    for k in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        strips[r].append(k)
        load[r] += float(strip_cost[k]) + 1e-9  # (all-zero costs still deal round-robin)
    tiles = []
    for r in range(world_size):
        strips[r].sort()
        ids = [t for k in strips[r] for row in range(k * strip_tile_rows, min((k + 1) * strip_tile_rows, ty)) for t in range(row * tx, (row + 1) * tx)]
        ids.sort(key=lambda t: (-cost.flat[t], t))
        tiles.append(np.asarray(ids, dtype=np.uint32))
    return strips, tiles


def heavy_tiles(costs_desc, resident_waves: int = 6144, trigger: float = 2.0, ratio: float = 0.3) -> int:
    """How many of a share's tiles (measured costs = longest wavefront per tile in shader cycles, sorted heaviest first) are worth two lanes per ray
    (PlanetAtmosphere.render_tiles_prepared(n_heavy=...), atmo_render_tiles_split): the library's own rule for whole frames (csrc/atmo_api.hip,
    heavy_tile_count) on exact costs.  The share's duration is estimated as the sum of its wave lifetimes (two waves per tile) over the waves the GPU
    holds (1024 SIMDs x 6); if the heaviest tile does not outlive `trigger` x that, the share is bound by throughput and nothing is split; otherwise the
    tiles outliving `ratio` x it are, at most a third of the share."""
    c = np.asarray(costs_desc, dtype=np.float64)
    if c.size == 0 or c.sum() <= 0.0:
        return 0
    draw = c.sum() * 2.0 / float(resident_waves)
    if not c[0] > trigger * draw:
        return 0
    return int(min(np.count_nonzero(c > ratio * draw), c.size // 3))


class StripGather:
    """Gathers the strips each rank drew into the frame on rank `dst`.

    Every rank renders into a full (H, W, 4) frame buffer of its own (atmo_render_tiles addresses the rect like atmo_render; only its
    tiles are written), packs its strips into a send buffer (one indexed copy), and the root scatters what it receives into the frame
    (one indexed copy per gather): strips are whole pixel rows, so both are contiguous row blocks.  Ranks hold different numbers of
    strips (LPT balances cost, not count): the send buffers are padded to the largest count."""

    def __init__(self, height: int, width: int, strips_of_rank, strip_rows: int, device, dst: int = 0, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world, self.dst = dist.get_rank(group), dist.get_world_size(group), dst
        if len(strips_of_rank) != self.world:
            raise ValueError("need one strip list per rank")
        self.h, self.w, self.strip_rows = height, width, strip_rows
        self.n_strips = (height + strip_rows - 1) // strip_rows
        flat = sorted(k for s in strips_of_rank for k in s)
        if flat != list(range(self.n_strips)):
            raise ValueError("the strip lists must partition the frame's strips")
        self.padded_h = self.n_strips * strip_rows
        self.max_strips = max(1, max(len(s) for s in strips_of_rank))
        self.mine = torch.as_tensor(list(strips_of_rank[self.rank]), dtype=torch.int64, device=device)
        self.frame = torch.zeros((self.padded_h, width, 4), dtype=torch.float32, device=device)  # this rank renders into frame[:height]
        self.send = torch.zeros((self.max_strips, strip_rows, width, 4), dtype=torch.float32, device=device)
        self.recv = None
        if self.rank == dst:
            self.recv = torch.empty((self.world, self.max_strips, strip_rows, width, 4), dtype=torch.float32, device=device)
            self.src_index = torch.as_tensor([r * self.max_strips + i for r, s in enumerate(strips_of_rank) for i in range(len(s))],
                                             dtype=torch.int64, device=device)
            self.dst_index = torch.as_tensor([k for s in strips_of_rank for k in s], dtype=torch.int64, device=device)

    def render_target(self):
        """(H, W, 4): where this rank's tile-list draw writes (rows beyond H exist only as padding of the last strip)."""
        return self.frame[: self.h]

    def gather(self):
        """Collective: on the root returns the assembled (H, W, 4) frame, elsewhere None."""
        torch = self.torch
        strips = self.frame.view(self.n_strips, self.strip_rows, self.w, 4)
        if self.mine.numel():
            self.send[: self.mine.numel()] = strips.index_select(0, self.mine)
        gather_list = [self.recv[k] for k in range(self.world)] if self.rank == self.dst else None
        self.dist.gather(self.send, gather_list, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        flat = self.recv.view(self.world * self.max_strips, self.strip_rows, self.w, 4)
        strips.index_copy_(0, self.dst_index, flat.index_select(0, self.src_index))
        return self.frame[: self.h]
