import sys, time
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from course5_amd import capi, meshgen as mg
xyz, cells, alpha, q = mg.workload("c3")
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
ctx.upload_grid(xyz, cells, alpha, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
ctx.set_option("stage_timing", 0)
out = torch.zeros((1800, 2400, 2), dtype=torch.float32, device="cuda:0")
def run(n=300):
    for _ in range(n): ctx.render_device(out.data_ptr())
    ctx.synchronize()
    return ctx.walk_kernel_ms(reset=True)[0]
for prec, k, pad in ((0,0,0),(1,0,0),(1,128,0),(1,1024,0),(1,0,4096),(1,0,8192),(1,128,0),(0,0,0)):
    ctx.set_option("precision", prec); ctx.set_option("steep_ratio", k); ctx.set_option("lds_pad", pad)
    run(100); ms = run(300)
    print(f"precision {prec} steep_ratio {k} lds_pad {pad}: walk {ms:.4f} ms", flush=True)
