import sys, os
sys.path.insert(0, '/root/repo')
import torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0); ctx.set_option("depth_split", 1)
for wl, res, rows in (("c3", (2400, 1800), (0, -1)), ("c3", (2400, 1800), (838, 124)), ("c2", (1200, 900), (0, -1))):
    xyz, c, a, q = mg.workload(wl)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_row_range(0, -1)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_row_range(*rows)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    imgs = []
    for er in (0, 1, 0, 1):
        ctx.set_option("entry_records", er)
        for _ in range(40): ctx.render()
        best = None
        for _ in range(10):
            img = ctx.render(); st = ctx.stats()
            if best is None or st["ms_total"] < best["ms_total"]: best = st
        imgs.append(img.copy())
        print(wl, res, rows, "entry_records", er, "records %.4f entries %.4f walk %.4f total %.4f S %d entries %d" % (best["ms_records"], best["ms_entries"], best["ms_walk"], best["ms_total"], best["segments"], best["entries"]), flush=True)
    print("  images bit-equal:", (imgs[0].view('uint32') == imgs[1].view('uint32')).all())
