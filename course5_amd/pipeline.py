"""Frame pipeline of the multi-GPU path: each rank renders its rows, one gather of the strips to
rank 0 per frame (RCCL on the GPU box; gloo in the CPU tests), reassembly there.

The reference has no exchange step (one process, plane.cpp:161-169 loops over all pixels); the
gather exists only because the image plane is sharded.  Up to `depth` gathers stay in flight so
the exchange of frame k overlaps the render of frame k + 1.

Two row layouts:
  cyclic  tiles of `tile_rows` rows dealt round-robin (c5_set_row_tiles): balanced by
          construction, but every rank touches the whole grid, so the per-view setup is replicated;
  blocks  one contiguous block per rank (c5_set_row_range), sized by measured per-row cost
          (sharding.balanced_blocks): each rank builds only the records its rays can reach, which
          shards the per-view setup as well.  A block is contiguous in the image, so rank 0 RECEIVES every
          block at its final offset (one irecv per rank straight into a view of the frame, its own block
          rendered in place): no gather into staging parts, no reassembly pass over the image.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import sharding


class FramePipeline:
    def __init__(self, res_x: int, res_y: int, rank: int, world: int, device, depth: int = 2,
                 tile_rows: int = 16, blocks=None, host_staging: bool = False, check=None):
        """host_staging: rehearsal only (gloo backend, which cannot gather device tensors): strips
        take a round trip through host memory around the gather.
        check: callable returning the status of the frame just rendered (Context.synchronize: C5_OK or
        C5_RETRY).  With it no strip is gathered before its render is known to be complete: a frame
        reported C5_RETRY (an internal buffer was too small) is rendered again by this rank alone, before
        its one gather of the step, so the ranks stay in step without exchanging flags.  The host then
        waits for each render, but the gather of frame k still overlaps the render of frame k + 1."""
        self.host_staging = host_staging
        self.check = check
        self.retries = 0
        self.res_x, self.res_y, self.tile_rows = res_x, res_y, tile_rows
        self.rank, self.world, self.device = rank, world, device
        self.blocks = list(blocks) if blocks is not None else None
        self.depth = max(1, depth) if world > 1 else 1
        if world == 1:
            self.pad_rows = res_y
        elif self.blocks is not None:
            self.pad_rows = max(n for _, n in self.blocks)
        else:
            self.pad_rows = sharding.padded_rows(res_y, tile_rows, world)
        self.strips = [torch.zeros((self.pad_rows, res_x, 2), dtype=torch.float32, device=device)
                       for _ in range(self.depth)]
        self.frame = None
        self.parts = None
        self.row_index = None
        self.frames = None
        # blocks, exchanged on the device (or between CPU tensors): point-to-point, received in place
        self.in_place = world > 1 and self.blocks is not None and not host_staging
        if world > 1 and rank == 0:
            if self.in_place:
                self.frames = [torch.empty((res_y, res_x, 2), dtype=torch.float32, device=device) for _ in range(self.depth)]
                self.frame = self.frames[0]
            else:
                self.parts = [[torch.empty_like(self.strips[0]) for _ in range(world)] for _ in range(self.depth)]
                self.frame = torch.empty((res_y, res_x, 2), dtype=torch.float32, device=device)
            if self.blocks is None:
                self.row_index = [torch.from_numpy(sharding.local_rows(res_y, tile_rows, r, world)).to(device)
                                  for r in range(world)]
        self.pending = []
        self.k = 0
        # Rank 0 reassembles the frame on a stream of its own, beside its next render (the copies are
        # bandwidth-bound, the walk is not): `assembled[s]` marks when parts[s] may be overwritten again.
        on_gpu = getattr(device, "type", str(device)) == "cuda"
        self.side = torch.cuda.Stream(device=device) if (world > 1 and rank == 0 and on_gpu) else None
        self.assembled = [None] * self.depth

    def _assemble(self, s):
        for r in range(self.world):
            if self.blocks is not None:
                b, n = self.blocks[r]
                self.frame[b:b + n].copy_(self.parts[s][r][:n])
            else:
                idx = self.row_index[r]
                self.frame.index_copy_(0, idx, self.parts[s][r][: idx.numel()])

    def _finish(self, slot):
        work, s = slot
        if self.in_place:
            work.wait()  # on GPU: the current stream waits for the transfers, the host does not
            if self.rank == 0:
                self.frame = self.frames[s]
            return
        if self.side is None:
            work.wait()  # on GPU: the current stream waits for the collective, the host does not
            if self.rank == 0:
                self._assemble(s)
            return
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)  # (host-staged rehearsal: the parts were copied in on the current stream)
        with torch.cuda.stream(self.side):
            work.wait()  # the side stream waits for the collective
            self._assemble(s)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.assembled[s] = ev

    def step(self, render):
        """render(strip) must fill strip[:local_rows] (enqueue on the current stream on GPU)."""
        s = self.k % self.depth
        self.k += 1
        if self.assembled[s] is not None:  # the gather of this step overwrites parts[s]
            torch.cuda.current_stream().wait_event(self.assembled[s])
            self.assembled[s] = None
        # blocks received in place: rank 0 renders its own block straight into the frame
        target = self.strips[s]
        if self.in_place and self.rank == 0:
            b0, n0 = self.blocks[0]
            target = self.frames[s][b0:b0 + n0]
        render(target)
        if self.check is not None:
            for _ in range(4):
                if self.check() == 0:
                    break
                self.retries += 1
                render(target)
            else:
                raise RuntimeError("a frame kept being reported incomplete (C5_RETRY)")
        if self.world == 1:
            self.frame = self.strips[s]
            return
        if self.in_place:
            if self.rank == 0:
                ops = [dist.P2POp(dist.irecv, self.frames[s][b:b + n], r) for r, (b, n) in enumerate(self.blocks) if r != 0 and n > 0]
            else:
                n = self.blocks[self.rank][1]
                ops = [dist.P2POp(dist.isend, self.strips[s][:n], 0)] if n > 0 else []
            work = _All(dist.batch_isend_irecv(ops)) if ops else _Done()
        elif self.host_staging:
            torch.cuda.current_stream().synchronize()
            host = self.strips[s].cpu()
            got = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, got, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.parts[s][r].copy_(got[r])
            work = _Done()
        else:
            work = dist.gather(self.strips[s], self.parts[s] if self.rank == 0 else None, dst=0, async_op=True)
        self.pending.append((work, s))
        if len(self.pending) >= self.depth:
            self._finish(self.pending.pop(0))

    def drain(self):
        while self.pending:
            self._finish(self.pending.pop(0))
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        return self.frame


class _Done:
    def wait(self):
        return True


class _All:
    """The transfers of one step (dist.batch_isend_irecv) as one thing to wait for."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()
        return True


def gather_row_costs(local_costs: np.ndarray, blocks, rank: int, world: int, device) -> np.ndarray:
    """All ranks: per-row costs of the whole image from each rank's block (one small all_gather)."""
    res_y = sum(n for _, n in blocks)
    if world == 1:
        return np.asarray(local_costs, dtype=np.int64)
    pad = max(n for _, n in blocks)
    mine = torch.zeros(pad, dtype=torch.int64, device=device)
    mine[: len(local_costs)] = torch.from_numpy(np.asarray(local_costs, dtype=np.int64)).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.zeros(res_y, dtype=np.int64)
    for (b, n), p in zip(blocks, parts):
        out[b:b + n] = p[:n].cpu().numpy()
    return out
