"""A/B of one option of the library in ONE process, runs interleaved; also says whether the images are bit-equal.
usage: option_probe.py NAME VALUE_A VALUE_B [rounds] [workload]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from course5_amd import capi, meshgen as mg  # noqa: E402

name, va, vb = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
workload = sys.argv[5] if len(sys.argv) > 5 else "c3"
res = {"c3": (2400, 1800), "c2": (1200, 900), "c3@1200": (1200, 900), "c3@4800": (4800, 3600)}[workload]
xyz, cells, alpha, q = mg.workload(workload.split("@")[0])
outs, ctxs = [], []
for v in (va, vb):
    ctx = capi.Context(0)
    ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
    ctx.upload_grid(xyz, cells, alpha, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
    ctx.set_option("stage_timing", 0)
    ctx.set_option(name, v)
    ctxs.append(ctx)
    outs.append(torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0"))


def run(k, n):
    for _ in range(n):
        ctxs[k].render_device(outs[k].data_ptr())
    ctxs[k].synchronize()
    return ctxs[k].walk_kernel_ms(reset=True)[0]


for k in range(2):
    run(k, 300)
same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
print("images bit-equal:", same, "| segments", ctxs[0].stats()["segments"], ctxs[1].stats()["segments"], flush=True)
tot = [0.0, 0.0]
for r in range(rounds):
    ms = [run(0, 300), run(1, 300)]
    tot = [tot[0] + ms[0], tot[1] + ms[1]]
    print("round %d  %s=%g %.4f | %s=%g %.4f" % (r, name, va, ms[0], name, vb, ms[1]), flush=True)
print("mean walk ms: %s=%g %.4f | %s=%g %.4f" % (name, va, tot[0] / rounds, name, vb, tot[1] / rounds))
