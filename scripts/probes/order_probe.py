"""Scratch probe: the order the rows of super-blocks start in - by the rows' segment sums (library) or by their longest ray
(C5_LIB=..._sbmax.so), for small frames only ("cost_order" 1) or for every frame (2), against image order (0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
def timed():
    for _ in range(30):
        ctx.render()
    w = sorted((ctx.render() is not None and ctx.stats()["ms_walk"]) for _ in range(12))
    return w[0], w[6]
for wl, res, shares in (("c3", (2400, 1800), ((0, -1), (0, 900), (900, 900), (0, 652), (652, 248), (1148, 652))),
                        ("c3", (4800, 3600), ((0, 1028), (1028, 276), (1676, 248), (2772, 1028))),
                        ("c2", (1200, 900), ((0, -1),)), ("c2", (600, 450), ((0, -1),))):
    xyz, c, a, q = mg.workload(wl)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    ctx.set_row_range(0, -1)
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    for rows in shares:
        ctx.set_row_range(0, -1); ctx.set_row_range(*rows)
        out = []
        for co in (0, 1, 2):
            ctx.set_option("cost_order", co)
            out.append("%.4f/%.4f" % timed())
        print(wl, res, rows, "walk ms best/median  image order:", out[0], " small frames:", out[1], " always:", out[2], flush=True)
ctx.set_option("cost_order", 1)
