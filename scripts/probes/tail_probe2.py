"""Scratch probe: the SPLIT instantiation with (nearly) every tile whole against the whole-ray kernel."""
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
    for _ in range(30):
        ctx.render()
    w = []
    for _ in range(20):
        ctx.render()
        st = ctx.stats()
        w.append((st["ms_walk"], st["ms_total"], st["steps"]))
    w.sort()
    return w[0], w[10]
for k, tail in ((1, 1.0), (2, 0.0005), (2, 0.02), (2, 0.05), (3, 0.02), (3, 0.05), (1, 1.0)):
    ctx.set_option("depth_split", k)
    ctx.set_option("split_tail", tail)
    print(k, tail, timed(), flush=True)
