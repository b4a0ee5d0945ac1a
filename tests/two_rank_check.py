"""Rehearsal of the N = 2 frame pipeline on ONE GPU (both ranks on device 0, gloo, strips staged through
the host): every frame rank 0 reassembles must equal the single-context render of the same view.

    C5_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29533 tests/two_rank_check.py [cyclic|blocks]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg, sharding  # noqa: E402
from course5_amd.pipeline import FramePipeline  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "cyclic"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
res_x, res_y, tile_rows = 640, 480, 16
xyz, cells, alpha, q = mg.workload("c2")
views = [mg.view_rotations(0.1 + 0.05 * k, 0.07 - 0.04 * k) for k in range(7)]

ctx = capi.Context(0)
if os.environ.get("C5_PIPELINE"):  # two frame slots inside the context as well
    ctx.set_option("pipeline", int(os.environ["C5_PIPELINE"]))
ctx.set_option("depth_split", 1)  # the reassembled frames are compared bit for bit with single-context renders: whole rays
ctx.upload_grid(xyz, cells, alpha, q)
ctx.set_image(res_x, res_y, mg.REFERENCE_BOUNDS)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
blocks = None
if layout == "cyclic":
    ctx.set_row_tiles(tile_rows, rank, world)
else:
    blocks = sharding.equal_blocks(res_y, world)
    ctx.set_row_range(*blocks[rank])
if os.environ.get("C5_ENTRY_POOL"):  # start from a pool that is too small: frames must be re-rendered, never wrong
    ctx.set_option("entry_pool", int(os.environ["C5_ENTRY_POOL"]))
# no strip is gathered before its render is known to be complete (C5_RETRY -> this rank renders again)
pipe = FramePipeline(res_x, res_y, rank, world, dev, depth=2, tile_rows=tile_rows, blocks=blocks, host_staging=True,
                     check=ctx.synchronize)
got = []


def step(v):
    def render(strip):
        ctx.set_view(v)
        ctx.render_device(strip.data_ptr())
    pipe.step(render)


with torch.cuda.stream(stream):
    # frame by frame, drained each time, so that every reassembled frame can be compared
    for v in views:
        step(v)
        frame = pipe.drain()
        torch.cuda.synchronize()
        if rank == 0:
            got.append(frame.cpu().numpy().copy())
    # and back to back (two gathers in flight), comparing the last frame only
    if os.environ.get("C5_ENTRY_POOL"):
        ctx.set_option("entry_pool", int(os.environ["C5_ENTRY_POOL"]))
    for v in views:
        step(v)
    last = pipe.drain()
    torch.cuda.synchronize()
    assert ctx.synchronize() == capi.C5_OK
    print(f"rank {rank}: {pipe.retries} frame(s) rendered again after C5_RETRY", flush=True)
ok = True
if rank == 0:
    full = capi.Context(0)
    full.set_option("depth_split", 1)
    full.upload_grid(xyz, cells, alpha, q)
    full.set_image(res_x, res_y, mg.REFERENCE_BOUNDS)
    from oracle.pyoracle import Oracle  # the checker: the reassembled frames also against the CPU oracle
    oracle = Oracle("port")
    for k, v in enumerate(views):
        full.set_view(v)
        want = full.render()
        same = np.array_equal(got[k].view(np.uint32), want.view(np.uint32))
        ref = oracle.render(xyz, cells, alpha, q, v, res_x, res_y, mg.REFERENCE_BOUNDS, threads=4)["image"]
        a, b = got[k].astype(np.float64), ref.astype(np.float64)
        close = bool((np.abs(a - b) <= 1e-5 * np.maximum(np.abs(a), np.abs(b)) + 1e-6 * np.abs(b).max()).all())
        ok &= same and close
        print(f"{layout} frame {k}: {'equal' if same else 'DIFFERENT'} to the single-context render, "
              f"{'within' if close else 'BEYOND'} 1e-5 of the oracle", flush=True)
    same = np.array_equal(last.cpu().numpy().view(np.uint32), want.view(np.uint32))
    ok &= same
    print(f"{layout} last frame of the back-to-back burst: {'equal' if same else 'DIFFERENT'}", flush=True)
    print("PASS" if ok else "FAIL", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
