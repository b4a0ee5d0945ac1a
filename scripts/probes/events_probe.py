"""Scratch probe: what a frame's events cost (frames back to back, wall clock): the six stage events, the two around the walk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_view(mg.view_rotations(0.1, 0.07))
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
out = torch.zeros((1800, 2400, 2), dtype=torch.float32, device="cuda:0")
def wall(n=400):
    for _ in range(100): ctx.render_device(out.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): ctx.render_device(out.data_ptr())
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
res = {}
for st, wt in ((0, 0), (0, 1), (1, 0), (0, 0), (0, 1), (1, 0)):
    ctx.set_option("stage_timing", st); ctx.set_option("walk_timing", wt)
    res.setdefault((st, wt), []).append(wall())
for k, v in res.items():
    print("stage_timing %d walk_timing %d:" % k, " ".join("%.4f" % x for x in v))
