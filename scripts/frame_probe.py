"""A/B of whole frames (all kernels of a frame, image left in HBM) of two or more builds in ONE process, runs
interleaved.  usage: frame_probe.py libA.so libB.so [...] [rounds] [workload]"""
import os
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from course5_amd import capi, meshgen as mg  # noqa: E402

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
rounds = int(rest[0]) if len(rest) > 0 else 4
workload = rest[1] if len(rest) > 1 else "c3"
res = {"c3": (2400, 1800), "c2": (1200, 900), "c3@1200": (1200, 900), "c3@4800": (4800, 3600)}[workload]
xyz, cells, alpha, q = mg.workload(workload.split("@")[0])
out = torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0")
ctxs = []
for p in paths:
    capi._lib = None
    capi.LIB_PATH = p
    ctx = capi.Context(0)
    ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
    ctx.upload_grid(xyz, cells, alpha, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
    ctx.set_option("stage_timing", 0)
    for kv in os.environ.get("C5_OPTS", "").split(","):
        if kv:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    ctxs.append(ctx)


def run(ctx, n):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


for ctx in ctxs:
    run(ctx, 300)
tot = [0.0] * len(paths)
for r in range(rounds):
    line = []
    for k, ctx in enumerate(ctxs):
        ms = run(ctx, 500)
        tot[k] += ms
        line.append("%s %.4f" % (paths[k].split("/")[-1], ms))
    print("round", r, " | ".join(line), flush=True)
print("mean ms per frame:", " | ".join("%s %.4f" % (paths[k].split("/")[-1], tot[k] / rounds) for k in range(len(paths))))
