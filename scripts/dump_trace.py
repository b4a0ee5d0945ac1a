import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
lib = capi.load_library()
n_blocks = 131072
buf = (C.c_ulonglong * (4 * n_blocks))()
out = torch.zeros((1800, 2400, 2), dtype=torch.float32, device="cuda:0")
for _ in range(200): ctx.render_device(out.data_ptr())
ctx.synchronize()
lib.c5_debug_walk_trace(buf, n_blocks, 1)
ctx.render(); st = ctx.stats()
lib.c5_debug_walk_trace(buf, n_blocks, 1)
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
np.save("/root/repo/gpurun_out/walk_trace.npy", t)
print("saved", int((t[:, 1] > 0).sum()), "walk ms", st["ms_walk"])
