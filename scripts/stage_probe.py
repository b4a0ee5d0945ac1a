"""Per-stage GPU times of a frame (HIP events per stage: c5_stats), sustained.  C5_LIB selects the build."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from course5_amd import capi, meshgen as mg  # noqa: E402
for name, res in (("c3", (2400, 1800)), ("c2", (1200, 900))):
    xyz, cells, alpha, q = mg.workload(name)
    ctx = capi.Context(0)
    ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
    ctx.upload_grid(xyz, cells, alpha, q)
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
    out = torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0")
    acc = {}
    for k in range(260):
        ctx.render_device(out.data_ptr())
        st = ctx.stats()
        if k >= 60:
            for key in ("ms_transform", "ms_records", "ms_entries", "ms_solids", "ms_walk", "ms_total"):
                acc[key] = acc.get(key, 0.0) + st[key] / 200
    print(capi.LIB_PATH.split("/")[-1], name, {k: round(v, 4) for k, v in acc.items()}, flush=True)
    ctx.close()
