"""Scratch: sustained C3 frame time against the height of the XCD bands of the walk."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
res = (2400, 1800) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
ctx.set_option("stage_timing", 0)
out = torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0")
for xm, rows in ((0, 0), (1, 16), (2, 32), (2, 64), (2, 128), (2, 256)):
    ctx.set_option("xcd_mode", xm)
    ctx.set_option("band_rows", rows)
    for _ in range(30):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    ctx.walk_kernel_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 200 * 1e3
    ms, n = ctx.walk_kernel_ms(reset=True)
    print("xcd_mode", xm, "band_rows", rows, "frame ms", round(dt, 4), "walk ms", round(ms, 4), flush=True)
