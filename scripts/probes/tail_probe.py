"""Scratch probe: only the tail of the walk's launch cut in slabs ("split_tail") against whole rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_view(mg.view_rotations(0.1, 0.07))

def timed():
    for _ in range(30):
        ctx.render()
    best = None
    for _ in range(10):
        img = ctx.render()
        st = ctx.stats()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    return best, img

for res, shares in (((2400, 1800), ((0, -1),)), ((4800, 3600), ((1676, 248), (0, -1)))):
    ctx.set_row_range(0, -1)
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    for rows in shares:
        ctx.set_row_range(*rows)
        ctx.set_option("depth_split", 1)
        ref, img0 = timed()
        print(res, rows, "whole: setup %.4f walk %.4f frame %.4f" % (ref["ms_transform"] + ref["ms_records"] + ref["ms_entries"], ref["ms_walk"], ref["ms_total"]), flush=True)
        for k in (2, 3, 4):
            for tail in (0.1, 0.2, 0.3, 0.45):
                ctx.set_option("depth_split", k)
                ctx.set_option("split_tail", tail)
                st, img = timed()
                ok = st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered_pixels"]
                d = np.abs(img.astype(np.float64) - img0).max()
                print(res, rows, "K %d tail %.2f: setup %.4f walk %.4f frame %.4f  steps +%d  %s maxdiff %.2e" % (
                    k, tail, st["ms_transform"] + st["ms_records"] + st["ms_entries"], st["ms_walk"], st["ms_total"], st["steps"] - ref["steps"], "ok" if ok else "COUNTS DIFFER", d), flush=True)
        ctx.set_option("split_tail", 1.0)
