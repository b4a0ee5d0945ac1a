"""Per-stage GPU times (HIP events inside the library, option "stage_timing") of the C3 frame for the library named
by C5_LIB: mean over frames after a warm-up."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
from course5_amd import capi, meshgen as mg
xyz, cells, alpha, q = mg.workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
ctx.upload_grid(xyz, cells, alpha, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
acc = {}
for k in range(260):
    ctx.render()
    st = ctx.stats()
    if k >= 60:
        for key in ("ms_transform", "ms_records", "ms_entries", "ms_walk", "ms_total"):
            acc[key] = acc.get(key, 0.0) + st[key] / 200
print(capi.LIB_PATH.split("/")[-1], {k: round(v, 4) for k, v in acc.items()})
