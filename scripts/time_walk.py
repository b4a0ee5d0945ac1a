"""Scratch: best-of-N stage timings of the C3 frame for the default kernel, per tile shape and integration order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
ctx.upload_grid(xyz, c, a, q)
res = (2400, 1800) if len(sys.argv) < 3 else tuple(int(v) for v in sys.argv[2].split("x"))
ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
for order in (0, 1):
    for tile in (0, 1, 2):
        ctx.set_option("integration", order); ctx.set_option("tile", tile)
        best = None
        for i in range(12):
            ctx.render()
            st = ctx.stats()
            if best is None or st["ms_walk"] < best["ms_walk"]:
                best = st
        print("order", order, "tile", tile, "walk", round(best["ms_walk"], 3), "total", round(best["ms_total"], 3),
              "tr/rec/ent", round(best["ms_transform"], 3), round(best["ms_records"], 3), round(best["ms_entries"], 3),
              "S", best["segments"], flush=True)
