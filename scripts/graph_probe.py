"""Does replaying a frame as a hipGraph beat six eager launches?  (DESIGN.md section 9, round-1 idea.)
The frame of c5_render_device is captured with torch.cuda.graph on the stream handed to the context; walk /
stage timing events are switched off for the capture.  C2 ball at 1200x900 (0.15 ms per frame) and C3."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from course5_amd import capi, meshgen as mg  # noqa: E402

dev = torch.device("cuda", 0)
for name, res in (("c2", (1200, 900)), ("c3", (2400, 1800))):
    xyz, cells, alpha, q = mg.workload(name)
    ctx = capi.Context(0)
    ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
    ctx.upload_grid(xyz, cells, alpha, q)
    ctx.set_image(*res, mg.REFERENCE_BOUNDS)
    ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
    ctx.set_option("stage_timing", 0)
    ctx.set_option("walk_timing", 0)
    out = torch.zeros((res[1], res[0], 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        for _ in range(50):
            ctx.render_device(out.data_ptr())
        assert ctx.synchronize() == capi.C5_OK
        want = out.clone()

        def timed(fn, n=500):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / n

        eager = timed(lambda: ctx.render_device(out.data_ptr()))
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=stream):
                ctx.render_device(out.data_ptr())
            out.zero_()
            replay = timed(g.replay)
            same = bool(torch.equal(out, want))
            print(f"{name} {res[0]}x{res[1]}: eager {eager:.4f} ms/frame, graph replay {replay:.4f} ms/frame, image equal: {same}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name}: capture failed: {e!r}", flush=True)
    del g
    ctx.close()
