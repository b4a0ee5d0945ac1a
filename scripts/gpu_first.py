"""Scratch: first GPU run — parity vs oracle on small grids, timing on C3."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from course5_amd import meshgen as mg, capi
from oracle.pyoracle import Oracle

def cmp(a, b):
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b)
    a = np.where(nan_a, 0, a).astype(np.float64); b = np.where(nan_b, 0, b).astype(np.float64)
    out = {}
    for ch in range(2):
        A, B = a[..., ch], b[..., ch]
        tol = 1e-5 * np.maximum(np.abs(A), np.abs(B)) + 1e-6 * np.abs(B).max()
        bad = np.abs(A - B) > tol
        out[ch] = (int(bad.sum()), float(np.abs(A - B).max()), int((A != B).sum()))
    return out

o = Oracle("port")
ctx = capi.Context(0)
for name, res in (("c1", (600, 450)), ("g2", (600, 450)), ("g2", (120, 90)), ("c2", (300, 225))):
    xyz, cells, a, q = mg.workload(name)
    ctx.upload_grid(xyz, cells, a, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    for view in ((0.1, 0.07), (0.5, 0.25), (0.0, 0.0)):
        rots = mg.view_rotations(*view)
        ctx.set_view(rots)
        for tile in (0, 1, 2):
            ctx.set_option("tile", tile)
            img = ctx.render()
            st = ctx.stats()
            ref = o.render(xyz, cells, a, q, rots, res[0], res[1], mg.REFERENCE_BOUNDS, threads=8)
            print(name, res, view, "tile", tile, "S gpu/oracle", st["segments"], ref["segments"],
                  "cov", st["covered_pixels"], ref["covered"], "steps", st["steps"], cmp(img, ref["image"]), flush=True)

xyz, cells, a, q = mg.workload("c3")
t = time.time(); ctx.upload_grid(xyz, cells, a, q); print("upload c3 s", time.time() - t, flush=True)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
for tile in (0, 1, 2):
    for xm in (0, 1):
        ctx.set_option("tile", tile); ctx.set_option("xcd_mode", xm)
        img = ctx.render()
        img = ctx.render()
        st = ctx.stats()
        print("c3 tile", tile, "xcd", xm, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
print("Mrays/s", 2400 * 1800 / st["ms_total"] / 1e3)
