"""Scratch probe: walk time of the C3 frame and of shares of it, "cost_order" 0 / 1 / 2, interleaved (C5_LIB variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
def timed():
    for _ in range(100):
        ctx.render()
    w = sorted((ctx.render() is not None and ctx.stats()["ms_walk"]) for _ in range(16))
    return w[0], w[8]
for wl, res, shares in (("c3", (2400, 1800), ((0, -1), (0, 900), (900, 900), (0, 652), (652, 248), (838, 124))),
                        ("c3", (4800, 3600), ((0, 1028), (1028, 276), (1676, 248))),
                        ("c2", (1200, 900), ((0, -1),)), ("c2", (600, 450), ((0, -1),))):
    xyz, c, a, q = mg.workload(wl)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    ctx.set_row_range(0, -1)
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    for rows in shares:
        ctx.set_row_range(0, -1); ctx.set_row_range(*rows)
        out = {0: [], 1: [], 2: []}
        for co in (0, 1, 2, 2, 1, 0, 0, 1, 2):
            ctx.set_option("cost_order", co)
            out[co].append(timed())
        st = ctx.stats()
        print(os.environ.get("C5_LIB", "library")[-12:], wl, res, rows, "covered", st["covered_pixels"], "median walk ms:",
              " | ".join(f"order {co}: " + " ".join("%.4f" % m for _, m in out[co]) for co in (0, 1, 2)), flush=True)
ctx.set_option("cost_order", 1)
