"""Scratch: stress of "pipeline" 1 (two frame slots): bursts of asynchronous frames with changing views, with
and without a row range / row tiles, each frame compared with the serial render of the same view."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from course5_amd import capi, meshgen as mg

xyz, cells, alpha, q = mg.workload("c2")
views = [mg.view_rotations(0.1 + 0.05 * k, 0.07 - 0.04 * k) for k in range(7)]
res_x, res_y = 640, 480
bad = 0
for layout in ("full", "range", "tiles"):
    serial, piped = capi.Context(0), capi.Context(0)
    serial.set_option("view_cache", 0)
    piped.set_option("view_cache", 0)
    piped.set_option("pipeline", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for c in (serial, piped):
        c.upload_grid(xyz, cells, alpha, q)
        c.set_image(res_x, res_y, mg.REFERENCE_BOUNDS)
        if layout == "range":
            c.set_row_range(0, res_y // 2)
        elif layout == "tiles":
            c.set_row_tiles(16, 0, 2)
    rows = piped.local_rows
    want = []
    for v in views:
        serial.set_view(v)
        want.append(serial.render())
    outs = [torch.zeros((rows, res_x, 2), dtype=torch.float32, device="cuda:0") for _ in views]
    for rep in range(40):
        for o in outs:
            o.fill_(-1.0)
        torch.cuda.synchronize()
        for v, o in zip(views, outs):
            piped.set_view(v)
            piped.render_device(o.data_ptr())
        rc = piped.synchronize()
        if rc != capi.C5_OK:
            continue
        for k, (o, w) in enumerate(zip(outs, want)):
            if not np.array_equal(o.cpu().numpy().view(np.uint32), w.view(np.uint32)):
                bad += 1
                d = (o.cpu().numpy() != w)
                print(f"{layout} rep {rep} frame {k}: {int(d.sum())} values differ, rows {np.unique(np.nonzero(d)[0])[:8]}", flush=True)
    print(layout, "done", flush=True)
    serial.close(); piped.close()
print("mismatching frames:", bad)
sys.exit(1 if bad else 0)
