"""Scratch: PCIe-inclusive frame rate through c5_render (host output buffer)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
ctx.set_option("stage_timing", 0)
for _ in range(5): ctx.render()
t = time.perf_counter()
n = 50
for _ in range(n): ctx.render()
dt = (time.perf_counter() - t) / n
print(f"c5_render (host buffer, pageable): {dt*1e3:.3f} ms/frame -> {2400*1800/dt/1e6:.0f} Mrays/s")
