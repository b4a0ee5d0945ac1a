"""Scratch: per-step latency of walk_composite on a nearly idle GPU (a few rows of the C3 frame)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
ctx.set_option("row_costs", 1)
for rows in (1, 4, 16, 64, 128, 256):
    for order in (0, 1):
        ctx.set_option("integration", order)
        ctx.set_row_range(900, rows)
        best = None
        for _ in range(6):
            img = ctx.render(); st = ctx.stats()
            if best is None or st["ms_walk"] < best["ms_walk"]: best = st
        max_steps = None
        print("rows", rows, "order", order, "walk ms", round(best["ms_walk"], 4), "segments", best["segments"],
              "covered", best["covered_pixels"], "avg steps/ray", round(best["segments"] / max(best["covered_pixels"], 1), 1),
              "us/step(avg ray)", round(best["ms_walk"] * 1e3 / max(best["segments"] / max(best["covered_pixels"], 1), 1), 3), flush=True)
