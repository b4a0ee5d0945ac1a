"""Frames per second of back-to-back c5_render_device calls under option sets given on the command line:

    python scripts/opt_probe.py "pipeline=0" "pipeline=1" "cost_order=0" ...      (C5_WORKLOAD, C5_RES as elsewhere)

Options that must be set before the grid is uploaded (pipeline) get a context of their own.  Wall clock over 600
frames after 300 of warm-up, two output images in turn; the walk's own time by HIP events beside it.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg  # noqa: E402

res = tuple(int(v) for v in os.environ.get("C5_RES", "2400x1800").split("x"))
xyz, c, a, q = mg.workload(os.environ.get("C5_WORKLOAD", "c3"))
outs = [torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0") for _ in range(2)]
for spec in sys.argv[1:] or ["pipeline=0"]:
    opts = dict(kv.split("=") for kv in spec.split(",") if kv)
    ctx = capi.Context(0)
    ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
    for k in ("pipeline",):
        if k in opts:
            ctx.set_option(k, float(opts.pop(k)))
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
    ctx.set_option("stage_timing", 0)
    for k, v in opts.items():
        ctx.set_option(k, float(v))
    for k in range(300):
        ctx.render_device(outs[k & 1].data_ptr())
    ctx.synchronize()
    ctx.walk_kernel_ms(reset=True)
    t0 = time.perf_counter()
    for k in range(600):
        ctx.render_device(outs[k & 1].data_ptr())
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{spec}: {dt / 600 * 1e3:.4f} ms per frame = {res[0] * res[1] * 600 / dt / 1e6:.0f} Mrays/s; walk {ctx.walk_kernel_ms(reset=True)[0]:.4f} ms (its own stream)", flush=True)
    del ctx
