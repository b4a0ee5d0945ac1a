"""Scratch: phase clock of walk_composite_lds (library built with -DC5_WALK_STAMPS=1, see walk_kernels.hip).

    C5_LIB=course5_amd/libcourse5_hip_stamps.so python scripts/stamp_walk.py
"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
lib = capi.load_library()
out = torch.zeros((1800, 2400, 2), dtype=torch.float32, device="cuda:0")
buf = (C.c_ulonglong * 16)()
for i in range(3):
    ctx.render()
lib.c5_debug_walk_stamps(buf, 1)
for tile in (0, 1, 2):
    ctx.set_option("tile", tile)
    ctx.render()
    ctx.render()
    lib.c5_debug_walk_stamps(buf, 1)
    ctx.render()
    lib.c5_debug_walk_stamps(buf, 1)
    w = list(buf)
    it = max(w[12], 1)
    print(f"tile {tile}: iterations {w[12]}  walking lanes/iter {w[13] / it:.1f}  runs/iter {w[10] / it:.1f}  distinct cells/iter {w[11] / it:.1f}")
names = ["election (lanes -> slots)", "ids read + staging loads issued", "emission step (exp)", "loads landed (+ ds_write)", "ds_read + geometry + exit"]
for stage in (1, 2):
    ctx.set_option("tile", 2)
    ctx.set_option("lds_stage", stage)
    for _ in range(300):  # sustained clocks (the first frames after an idle spell run slower)
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    lib.c5_debug_walk_stamps(buf, 1)
    ctx.render()
    st = ctx.stats()
    lib.c5_debug_walk_stamps(buf, 1)
    v = list(buf)
    tot = sum(v[:5])
    print("lds_stage", stage, "walk ms", st["ms_walk"], "lane-steps", st["steps"], "wavefronts", v[9], "loop ticks/wave", v[8] / max(v[9], 1),
          "ticks per wave-step", tot / max(v[12], 1))
    for n, x in zip(names, v[:5]):
        print(f"  {n:34s} {x:14d} ticks  {100.0 * x / tot:5.1f} %   {x / max(v[12], 1):7.2f} per wave-step")
