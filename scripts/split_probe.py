"""What "depth_split" buys: walk and frame times of whole frames and of one GPU's share of a frame with the rays whole and cut
in 2 / 3 / 4 slabs.  python scripts/split_probe.py [tag]  ->  profiles/<tag>_split_probe.md"""
import os
import sys

import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg  # noqa: E402
from course5_amd.build import kernel_source_hash  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # identical frames only because this is a benchmark: each does its whole per-view setup


def timed():
    for _ in range(30):
        ctx.render()
    best = None
    for _ in range(10):
        ctx.render()
        st = ctx.stats()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    return best


lines = [f"# {tag}: option \"depth_split\" — rays whole (1) against rays cut in K slabs; `python scripts/split_probe.py`", "",
         f"kernel sources {kernel_source_hash()}; best of 10 frames after 30 warm ones, HIP events per stage (stage timing on: a frame here is a few "
         "per cent slower than in bench.py); setup = transform + records (+ plane raster) + entries", "",
         "| workload | rows (first, count) | slabs | setup ms | walk ms | frame ms | steps | segments |", "|---|---|---|---|---|---|---|---|"]
for wl, res, rows_list in (("c3", (2400, 1800), [(0, -1), (838, 124), (776, 248), (0, 514)]),
                           ("c3", (1200, 900), [(0, -1)]),
                           ("c3", (4800, 3600), [(1676, 248), (1552, 496)]),
                           ("c2", (1200, 900), [(0, -1)])):
    xyz, c, a, q = mg.workload(wl)
    ctx.upload_grid(xyz, c, a, q)
    ctx.set_view(mg.view_rotations(0.1, 0.07))
    for rows in rows_list:
        ctx.set_row_range(0, -1)
        ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
        ctx.set_row_range(*rows)
        for k in (1, 2, 3, 4, 0):
            ctx.set_option("depth_split", k)
            st = timed()
            setup = st["ms_transform"] + st["ms_records"] + st["ms_entries"]
            name = "auto" if k == 0 else str(k)
            print(f"{wl} {res[0]}x{res[1]} rows {rows} slabs {name}: setup {setup:.4f} walk {st['ms_walk']:.4f} frame {st['ms_total']:.4f} steps {st['steps']}", flush=True)
            lines.append(f"| {wl} {res[0]}x{res[1]} | {rows} | {name} | {setup:.4f} | {st['ms_walk']:.4f} | {st['ms_total']:.4f} | {st['steps']} | {st['segments']} |")
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_split_probe.md"), "w").write("\n".join(lines) + "\n")
