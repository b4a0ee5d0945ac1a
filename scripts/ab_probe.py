"""A/B of two builds of the library in ONE process, runs interleaved (A B A B ...), so that clock and temperature
drift hits all alike.  usage: ab_probe.py libA.so libB.so [libC.so ...] [rounds] [workload]"""
import os
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from course5_amd import capi, meshgen as mg  # noqa: E402

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
rounds = int(rest[0]) if len(rest) > 0 else 5
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
    for kv in os.environ.get("C5_OPTS", "").split(","):  # e.g. C5_OPTS=lds_stage=2,band_rows=64
        if kv:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    ctxs.append(ctx)


def run(ctx, n):
    for _ in range(n):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    return ctx.walk_kernel_ms(reset=True)[0]


for ctx in ctxs:
    run(ctx, 300)
tot = [0.0] * len(paths)
for r in range(rounds):
    line = []
    for k, ctx in enumerate(ctxs):
        ms = run(ctx, 300)
        tot[k] += ms
        line.append("%s %.4f" % (paths[k].split("/")[-1], ms))
    print("round", r, " | ".join(line), flush=True)
print("mean walk ms:", " | ".join("%s %.4f" % (paths[k].split("/")[-1], tot[k] / rounds) for k in range(len(paths))))
