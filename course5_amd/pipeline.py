"""Frame pipeline of the multi-GPU path: each rank renders its row tiles, one gather of the
strips to rank 0 per frame (RCCL on the GPU box; gloo in the CPU tests), reassembly there.

The reference has no exchange step (one process, plane.cpp:161-169 loops over all pixels); the
gather exists only because the image plane is sharded.  Up to `depth` gathers stay in flight so
the exchange of frame k overlaps the render of frame k + 1.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import sharding


class FramePipeline:
    def __init__(self, res_x: int, res_y: int, tile_rows: int, rank: int, world: int, device, depth: int = 2):
        self.res_x, self.res_y, self.tile_rows = res_x, res_y, tile_rows
        self.rank, self.world, self.device = rank, world, device
        self.depth = max(1, depth) if world > 1 else 1
        self.pad_rows = sharding.padded_rows(res_y, tile_rows, world) if world > 1 else res_y
        self.strips = [torch.zeros((self.pad_rows, res_x, 2), dtype=torch.float32, device=device)
                       for _ in range(self.depth)]
        self.frame = None
        self.parts = None
        self.row_index = None
        if world > 1 and rank == 0:
            self.parts = [[torch.empty_like(self.strips[0]) for _ in range(world)] for _ in range(self.depth)]
            self.frame = torch.empty((res_y, res_x, 2), dtype=torch.float32, device=device)
            self.row_index = [torch.from_numpy(sharding.local_rows(res_y, tile_rows, r, world)).to(device)
                              for r in range(world)]
        self.pending = []
        self.k = 0

    def _finish(self, slot):
        work, s = slot
        work.wait()  # on GPU: the current stream waits for the collective, the host does not
        if self.rank == 0:
            for r in range(self.world):
                idx = self.row_index[r]
                self.frame.index_copy_(0, idx, self.parts[s][r][: idx.numel()])

    def step(self, render):
        """render(strip) must fill strip[:local_rows] (enqueue on the current stream on GPU)."""
        s = self.k % self.depth
        self.k += 1
        render(self.strips[s])
        if self.world == 1:
            self.frame = self.strips[s]
            return
        work = dist.gather(self.strips[s], self.parts[s] if self.rank == 0 else None, dst=0, async_op=True)
        self.pending.append((work, s))
        if len(self.pending) >= self.depth:
            self._finish(self.pending.pop(0))

    def drain(self):
        while self.pending:
            self._finish(self.pending.pop(0))
        return self.frame
