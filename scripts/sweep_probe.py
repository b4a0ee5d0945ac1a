"""A -Y sweep (the view turns a degree per frame, as utility/rotate_traces.py does) under option sets given on the command
line: mean walk time and frame time over 180 frames.  usage: sweep_probe.py "stage_slots=14" "stage_slots=21" ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg  # noqa: E402

res = tuple(int(v) for v in os.environ.get("C5_RES", "2400x1800").split("x"))
xyz, c, a, q = mg.workload(os.environ.get("C5_WORKLOAD", "c3"))
out = torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0")
views = [mg.view_rotations(mg.BENCH_VIEW["angle_around_x"], mg.BENCH_VIEW["angle_around_y"] + k / 180.0) for k in range(180)]
for spec in sys.argv[1:] or ["stage_slots=0"]:
    ctx = capi.Context(0)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_option("stage_timing", 0)
    for kv in spec.split(","):
        if kv:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    for rep in range(2):
        ctx.walk_kernel_ms(reset=True)
        t0 = time.perf_counter()
        for v in views:
            ctx.set_view(v)
            ctx.render_device(out.data_ptr())
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(f"{spec}: {dt / len(views) * 1e3:.4f} ms per frame, walk {ctx.walk_kernel_ms(reset=True)[0]:.4f} ms (mean over the sweep)", flush=True)
    del ctx
