"""Stage times of one GPU's share of the C3 frame (scratch probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_view(mg.view_rotations(0.1, 0.07))
for res, shares in (((2400, 1800), ((0, -1), (838, 124), (0, 514))), ((4800, 3600), ((0, -1), (1676, 248), (0, 1028)))):
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    for rows in shares:
        ctx.set_row_range(*rows)
        for _ in range(40):
            ctx.render()
        best = None
        for _ in range(10):
            ctx.render()
            st = ctx.stats()
            if best is None or st["ms_total"] < best["ms_total"]:
                best = st
        print(res, rows, {k: round(v, 4) for k, v in best.items() if k.startswith("ms_")}, "steps", best["steps"], flush=True)
