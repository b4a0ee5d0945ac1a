"""Scratch probe: option "cell_order" 0 / 1 (cells in the caller's order / in Morton order of their centroids): stage times
of the C3 frame and of shares of it, frame wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
xyz, c, a, q = mg.workload("c3")
out = torch.zeros((3600, 4800, 2), dtype=torch.float32, device="cuda:0")
def wall(n=300):
    ctx.set_option("stage_timing", 0)
    for _ in range(60):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    ctx.set_option("stage_timing", 1)
    return (time.perf_counter() - t0) * 1e3 / n
for order in (0, 1, 0, 1):
    ctx.set_option("cell_order", order)
    t0 = time.perf_counter(); ctx.upload_grid(xyz, c, a, q); t_up = time.perf_counter() - t0
    ctx.set_option("view_cache", 0)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    for res, shares in (((2400, 1800), ((0, -1), (838, 120), (0, 528))), ((4800, 3600), ((0, -1), (1676, 240), (0, 1064)))):
        ctx.set_row_range(0, -1)
        ctx.set_image(*res, mg.REFERENCE_BOUNDS)
        for rows in shares:
            ctx.set_row_range(*rows)
            for _ in range(60):
                ctx.render()
            best = None
            for _ in range(12):
                ctx.render()
                st = ctx.stats()
                if best is None or st["ms_total"] < best["ms_total"]:
                    best = st
            print("cell_order", order, "upload %.2f s" % t_up, res, rows, {k: round(v, 4) for k, v in best.items() if k in ("ms_records", "ms_entries", "ms_walk", "ms_total")}, "wall %.4f" % wall(), flush=True)
