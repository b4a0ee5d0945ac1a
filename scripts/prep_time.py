import sys, os, time
sys.path.insert(0, os.getcwd())
from course5_amd import capi, meshgen as mg
xyz, c, a, q = mg.workload("c3")
t0 = time.perf_counter(); ctx = capi.Context(0); t1 = time.perf_counter()
ctx.upload_grid(xyz, c, a, q); t2 = time.perf_counter()
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS); ctx.set_view(mg.view_rotations(0.1, 0.07)); t3 = time.perf_counter()
img = ctx.render(); t4 = time.perf_counter()
img = ctx.render(); t5 = time.perf_counter()
print(f"create {1e3*(t1-t0):.0f} ms, upload_grid {1e3*(t2-t1):.0f} ms, set_image+view {1e3*(t3-t2):.0f} ms, first render {1e3*(t4-t3):.0f} ms, second render {1e3*(t5-t4):.1f} ms")
ctx2 = capi.Context(0)
ctx2.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
t = time.perf_counter(); ctx2.upload_grid(xyz, c, a, q); print(f"second upload_grid {1e3*(time.perf_counter()-t):.0f} ms")
