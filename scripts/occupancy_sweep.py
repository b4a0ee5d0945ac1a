"""Scratch: C3 walk time against resident wavefronts per SIMD (extra dynamic LDS caps the workgroups per CU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
for pad_kb in (0, 8, 14, 22, 36, 62):
    ctx.set_option("lds_pad", pad_kb * 1024)
    best = None
    for i in range(10):
        ctx.render()
        st = ctx.stats()
        if best is None or st["ms_walk"] < best["ms_walk"]:
            best = st
    wg = min(6, int(160 * 1024 // (16896 + pad_kb * 1024)))  # 16.5 KB static LDS per workgroup; 80 VGPRs allow 6
    print("lds_pad KB", pad_kb, "workgroups/CU", wg, "walk", round(best["ms_walk"], 3), flush=True)
