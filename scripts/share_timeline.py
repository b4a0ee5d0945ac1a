"""Where the time of ONE GPU's share of a frame goes when its rays are cut in slabs (library built with -DC5_WALK_STAMPS=1):
every job's start and end (s_memrealtime) against the walk's HIP-event time and the frame's stage times.

    scripts/build_variant.sh stamps -DC5_WALK_STAMPS=1
    C5_LIB=course5_amd/libcourse5_hip_stamps.so python scripts/share_timeline.py [tag]   -> profiles/<tag>_share_timeline.md
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg  # noqa: E402
from course5_amd.build import kernel_source_hash  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
lib = capi.load_library()
n_blocks = 131072
buf = (C.c_ulonglong * (4 * n_blocks))()
lines = [f"# {tag}: one GPU's share of the C3 frame (2400x1800), job by job; kernel sources {kernel_source_hash()} (diagnostic build: the phase clock "
         "samples one job in 67)", "",
         "| rows (first, count) | slabs | walk ms (HIP events) | setup ms | jobs that walked | first start -> last end (us) | job life: median / 90 % / max (us) | "
         "steps per job: median / max | ns per wavefront-step | jobs started by (us): 50 % / 90 % / all | ended by: 50 % / 90 % |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for rows in ((838, 124), (0, 514)):
    ctx.set_row_range(0, -1)
    ctx.set_row_range(*rows)
    for k in (1, 2, 3, 4, 0):  # (0: the library's own choice - slabs from the statistics of the frame before, planes tilted to the view)
        ctx.set_option("depth_split", k)
        for _ in range(40):
            ctx.render()
        lib.c5_debug_walk_trace(buf, n_blocks, 1)
        ctx.render()
        st = ctx.stats()
        lib.c5_debug_walk_trace(buf, n_blocks, 1)
        t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
        t = t[(t[:, 1] > 0) & (t[:, 3] > 0)]
        b, e = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
        steps = t[:, 3].astype(np.int64)
        t0 = b.min()
        b, e = (b - t0) / 100.0, (e - t0) / 100.0
        life = e - b
        setup = st["ms_transform"] + st["ms_records"] + st["ms_entries"]
        row = (f"| {rows} | {k} | {st['ms_walk']:.4f} | {setup:.4f} | {len(t)} | {e.max():.1f} | {np.median(life):.1f} / {np.percentile(life, 90):.1f} / {life.max():.1f} | "
               f"{np.median(steps):.0f} / {steps.max()} | {np.median(life * 1e3 / np.maximum(steps, 1)):.0f} | "
               f"{np.percentile(b, 50):.1f} / {np.percentile(b, 90):.1f} / {b.max():.1f} | {np.percentile(e, 50):.1f} / {np.percentile(e, 90):.1f} |")
        print(row, flush=True)
        lines.append(row)
open(os.path.join(ROOT, "profiles", f"{tag}_share_timeline.md"), "w").write("\n".join(lines) + "\n")
