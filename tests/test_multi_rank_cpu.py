"""N > 1 path on CPU: two gloo ranks run the same FramePipeline bench.py uses on the GPU box
(row tiles dealt cyclically, one gather of strips per frame to rank 0, up to `depth` gathers in
flight, reassembly).  The per-rank renderer is the CPU oracle here (this container has no GPU);
on the GPU box the same sharding drives the HIP context
(tests/test_gpu_parity.py::test_c3_row_tile_shards_reassemble)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWS = [(0.1, 0.07), (0.5, 0.25), (0.0, 0.0), (1.3, -0.4), (0.2, 0.9)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tile_rows, depth, out_path):
    sys.path.insert(0, ROOT)
    from course5_amd import meshgen as mg, sharding
    from course5_amd.pipeline import FramePipeline, gather_row_costs
    from oracle.pyoracle import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xyz, cells, a, q = mg.workload("g2")
        rx, ry = 96, 70
        oracle = Oracle("port")
        if tile_rows > 0:   # cyclic row tiles
            rows = sharding.local_rows(ry, tile_rows, rank, world)
            pipe = FramePipeline(rx, ry, rank, world, torch.device("cpu"), depth=depth, tile_rows=tile_rows)
        else:               # contiguous blocks balanced by a cost measured on equal blocks first
            eq = sharding.equal_blocks(ry, world)
            probe = oracle.render(xyz, cells, a, q, mg.view_rotations(*VIEWS[0]), rx, ry, mg.REFERENCE_BOUNDS)["image"]
            b, n = eq[rank]
            local_cost = (probe[b:b + n, :, 0] > 0).sum(axis=1)  # stand-in for segments per row
            costs = gather_row_costs(local_cost, eq, rank, world, torch.device("cpu"))
            assert np.array_equal(costs, (probe[..., 0] > 0).sum(axis=1))
            blocks = sharding.balanced_blocks(costs, world, base_cost=1.0)
            b, n = blocks[rank]
            rows = np.arange(b, b + n)
            pipe = FramePipeline(rx, ry, rank, world, torch.device("cpu"), depth=depth, blocks=blocks)
        fulls, ok = [], []
        for k, view in enumerate(VIEWS):
            full = oracle.render(xyz, cells, a, q, mg.view_rotations(*view), rx, ry, mg.REFERENCE_BOUNDS)["image"]
            fulls.append(full)

            def render(strip, full=full):
                strip[: rows.size] = torch.from_numpy(full[rows])

            pipe.step(render)
            # with depth d the frame assembled after step k is frame k - d + 1
            done = k - (pipe.depth - 1)
            if rank == 0 and done >= 0:
                ok.append(np.array_equal(pipe.frame.numpy().view(np.uint32), fulls[done].view(np.uint32)))
        frame = pipe.drain()
        if rank == 0:
            ok.append(np.array_equal(frame.numpy().view(np.uint32), fulls[-1].view(np.uint32)))
            np.save(out_path, np.array(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows,depth", [(16, 1), (16, 2), (7, 3), (0, 2)],
                         ids=["cyclic16-d1", "cyclic16-d2", "cyclic7-d3", "balanced-blocks-d2"])
def test_two_ranks_gather_row_tiles(tmp_path, tile_rows, depth):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), tile_rows, depth, out), nprocs=2, join=True)
    ok = np.load(out)
    assert ok.size >= 2 and ok.all()
