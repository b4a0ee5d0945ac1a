"""Scratch probe: 14 against 21 staging slots on shares of the C3 frame that are mostly oblique-face tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_view(mg.view_rotations(0.1, 0.07))
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
def timed():
    for _ in range(100):
        ctx.render()
    w = sorted((ctx.render() is not None and ctx.stats()["ms_walk"]) for _ in range(16))
    return w[8]
for rows in ((0, -1), (0, 500), (0, 900), (652, 248), (1300, 500)):
    ctx.set_row_range(0, -1); ctx.set_row_range(*rows)
    out = {14: [], 21: []}
    for sl in (14, 21, 21, 14, 14, 21):
        ctx.set_option("stage_slots", sl)
        out[sl].append(timed())
    st = ctx.stats()
    print(rows, "steps", st["steps"], "median walk ms: 14 slots", " ".join("%.4f" % v for v in out[14]), "| 21 slots", " ".join("%.4f" % v for v in out[21]), flush=True)
ctx.set_option("stage_slots", 0)
