"""How close is "bit for bit"?  Prints, per golden fixture (made by the reference's object code) and view, how many fp32
values of the default fp64 walk differ from the reference's at all, and by how many fp32 ulps at most:

    python tests/report_differing.py            (needs a GPU; a checker like the tests beside it, not product code)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from course5_amd import capi  # noqa: E402
from parity import golden_fixtures, load_golden  # noqa: E402


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


ctx = capi.Context(0)
total = differing = 0
worst = 0
for path in golden_fixtures():
    fx = load_golden(path)
    rx, ry = (int(v) for v in fx["res"])
    stride = int(fx["stride"])
    ctx.upload_grid(fx["xyz"], fx["cells"], fx["alpha"], fx["q"])
    ctx.set_image(rx, ry, fx["bounds"])
    ctx.set_alpha_limit(float(fx["alpha_limit"]))
    for k in range(len(fx["views"])):
        ctx.set_view(fx[f"rots{k}"])
        img = np.ascontiguousarray(ctx.render()[::stride, ::stride])
        ref = np.ascontiguousarray(fx[f"image{k}"]).astype(np.float32)
        ok = ~np.isnan(ref)
        u = ulps(img[ok], ref[ok])
        n = int((u > 0).sum())
        total += int(ok.sum())
        differing += n
        worst = max(worst, int(u.max()) if u.size else 0)
        print(f"{fx['name']:<34} view {k}: {n:6d} of {int(ok.sum()):8d} values differ, at most {int(u.max()) if u.size else 0} ulp", flush=True)
print(f"all fixtures: {differing} of {total} fp32 values differ ({100.0 * differing / max(total, 1):.4f} %), at most {worst} ulp")
