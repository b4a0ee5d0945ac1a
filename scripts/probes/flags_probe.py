"""Scratch probe: option "tile_flags" 0 / 1, walk kernel ms and frame wall ms, interleaved, several workloads; images bit-equal?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
out = torch.zeros((3600, 4800, 2), dtype=torch.float32, device="cuda:0")
def run(n=300):
    ctx.set_option("stage_timing", 0)
    for _ in range(60):
        ctx.render_device(out.data_ptr())
    ctx.synchronize(); ctx.walk_kernel_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    w = (time.perf_counter() - t0) * 1e3 / n
    k = ctx.walk_kernel_ms(reset=True)[0]
    ctx.set_option("stage_timing", 1)
    return k, w
for wl, res, rows in (("c3", (2400, 1800), (0, -1)), ("c3", (4800, 3600), (0, -1)), ("c2", (1200, 900), (0, -1)), ("c3", (1200, 900), (0, -1)), ("c3", (2400, 1800), (838, 120)), ("c3", (4800, 3600), (1676, 240))):
    xyz, c, a, q = mg.workload(wl)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    ctx.set_row_range(0, -1)
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    ctx.set_row_range(*rows)
    imgs = {}
    res_ = {0: [], 1: []}
    for f in (1, 0, 0, 1, 1, 0):
        ctx.set_option("tile_flags", f)
        for _ in range(4): img = ctx.render()
        imgs[f] = img; st = ctx.stats()
        res_[f].append(run())
    same = np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    print(wl, res, rows, "bit-equal" if same else "IMAGES DIFFER", "| flags on: walk/frame", " ".join("%.4f/%.4f" % v for v in res_[1]), "| off:", " ".join("%.4f/%.4f" % v for v in res_[0]), flush=True)
ctx.set_option("tile_flags", 1)
