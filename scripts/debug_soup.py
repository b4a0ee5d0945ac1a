import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
from oracle.pyoracle import Oracle
rots = mg.view_rotations(0.13, 0.21)
xa, ca = mg.kuhn_box(3, lo=(0.6, -0.4, -0.3), size=0.6, jitter=0.1, seed=5)
xb, cb = mg.kuhn_box(4, lo=(0.85, -0.2, -0.45), size=0.7, jitter=0.1, seed=6)
xyz2 = np.vstack([xa, xb]); cells2 = np.vstack([ca, cb + len(xa)]).astype(np.int32)
a2, q2 = mg.scalars(len(cells2), seed=9)
ctx = capi.Context(0); ctx.upload_grid(xyz2, cells2, a2, q2); ctx.set_image(240, 180, mg.REFERENCE_BOUNDS); ctx.set_view(rots); ctx.set_option("algorithm", 1)
img = ctx.render(); st = ctx.stats(); print(st)
o = Oracle("port")
bad = None
ref = o.render(xyz2, cells2, a2, q2, rots, 240, 180, mg.REFERENCE_BOUNDS, threads=1)
d = np.argwhere(img != ref["image"])
print(len(d), d[:10])
probes = [(int(c), int(r)) for r, c, ch in d[:3]]
ref2 = o.render(xyz2, cells2, a2, q2, rots, 240, 180, mg.REFERENCE_BOUNDS, threads=1, probes=probes)
for (c, r), pr in zip(probes, ref2["probes"]):
    print("pixel", c, r, "gpu", img[r, c], "oracle", ref["image"][r, c], "n", len(pr))
    z = pr[:, 1]; print("  z_hi diffs min", np.min(np.abs(np.diff(z))) if len(z) > 1 else None, "ties", int((np.diff(z) == 0).sum()))
    for row in pr: print("   ", int(row[0]), repr(row[1]), repr(row[2]))
