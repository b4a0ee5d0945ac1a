"""Scratch: do two walks overlap usefully?  K frames on ONE context against K frames dealt alternately to TWO (three)
contexts of the same GPU (own streams, own per-view buffers): the tail of one frame's walk (a quarter of the launch at
falling occupancy, profiles/r03_walk_timeline.md) then overlaps the head of the next."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg

res = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2400x1800").split("x"))
K = 600
xyz, c, a, q = mg.workload("c3")
ctxs = []
for _ in range(3):
    ctx = capi.Context(0)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    ctx.set_option("stage_timing", 0)
    ctx.set_option("walk_timing", 0)
    ctxs.append(ctx)
outs = [torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0") for _ in range(3)]
for n in (1, 2, 3, 1, 2, 3):
    for k in range(100):
        ctxs[k % n].render_device(outs[k % n].data_ptr())
    for ctx in ctxs:
        ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ctxs[k % n].render_device(outs[k % n].data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for ctx in ctxs:
        assert ctx.synchronize() == 0
    print(f"{n} context(s): {dt * 1e3 / K:.4f} ms per frame = {res[0] * res[1] * K / dt / 1e6:.0f} Mrays/s", flush=True)
