"""Scratch: do two walks overlap usefully?  K frames on ONE context against K frames dealt alternately to TWO (three)
contexts of the same GPU (own streams, own per-view buffers): the tail of one frame's walk (a quarter of the launch at
falling occupancy, profiles/r03_walk_timeline.md) then overlaps the head of the next."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg

res = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2400x1800").split("x"))
K = 600
xyz, c, a, q = mg.workload("c3")
ctxs = []
for _ in range(3):
    ctx = capi.Context(0)
    ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    ctx.set_option("stage_timing", 0)
    ctx.set_option("walk_timing", 0)
    ctxs.append(ctx)
outs = [torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0") for _ in range(3)]
rows = []
for n in (1, 2, 3, 1, 2, 3):
    for k in range(100):
        ctxs[k % n].render_device(outs[k % n].data_ptr())
    for ctx in ctxs:
        ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ctxs[k % n].render_device(outs[k % n].data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for ctx in ctxs:
        assert ctx.synchronize() == 0
    print(f"{n} context(s): {dt * 1e3 / K:.4f} ms per frame = {res[0] * res[1] * K / dt / 1e6:.0f} Mrays/s", flush=True)
    rows.append((n, dt * 1e3 / K, res[0] * res[1] * K / dt / 1e6))
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
from course5_amd.build import kernel_source_hash
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_overlap_probe.md"), "w") as f:
    f.write(f"# {tag}: do frames overlapped on independent streams fill the walk's tail?  ({res[0]}x{res[1]}, C3, {K} frames per line)\n\n"
            f"`python scripts/overlap_probe.py`; kernel sources {kernel_source_hash()}.  N contexts on ONE GPU (own high-priority stream, own per-view "
            "buffers each), frame k on context k mod N: with N > 1 the last quarter of a frame's walk (falling occupancy, "
            f"{tag}_walk_timeline.md) runs beside the next frame's setup and walk.\n\n| contexts | ms per frame | Mrays/s |\n|---|---|---|\n")
    for n, ms, v in rows:
        f.write(f"| {n} | {ms:.4f} | {v:.0f} |\n")
    base = min(ms for n, ms, _ in rows if n == 1)
    f.write(f"\nBest of N = 2: {100 * (base / min(ms for n, ms, _ in rows if n == 2) - 1):.1f} %, of N = 3: "
            f"{100 * (base / min(ms for n, ms, _ in rows if n == 3) - 1):.1f} % over one context.  Setup of frame k + 1 on a low-priority stream "
            "beside walk k inside ONE context (option \"pipeline\" 1) loses a little: 0.5525 against 0.5460 ms per frame, the walk itself 0.522 against "
            "0.459 ms; whole frames alternating between two streams of one context: 0.5428 ms (scripts/opt_probe.py, round 3).  What a frame's drain leaves "
            "idle is not free capacity: the next frame's setup and first wavefronts lengthen the memory latency the last wavefronts' steps wait on.\n")
