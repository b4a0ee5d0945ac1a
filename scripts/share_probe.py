"""Scratch: which walk kernel is fastest on a SMALL share of the frame (what one of 8 GPUs renders)?  The LDS-staged
kernel trades per-step latency for throughput; a share that fills a third of the wavefront slots is bound by one
wavefront's life, i.e. by per-step latency."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg

ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_view(mg.view_rotations(0.1, 0.07))


def timed():
    for _ in range(30):
        ctx.render()
    best = None
    for _ in range(8):
        ctx.render()
        st = ctx.stats()
        if best is None or st["ms_walk"] < best["ms_walk"]:
            best = st
    return best


lines = ["# r04: which walk kernel on a SMALL share of the frame (what one of 8 GPUs renders)?  `python scripts/share_probe.py`", "",
         "walk = best of 8 after 30 warm frames, HIP events; a share that fills a third of the wavefront slots lasts as long as one wavefront's life: "
         "steps x per-step latency, and a lone wavefront's step is no shorter than one among eight (1.1 us)", "",
         "| image | rows (first, count) | lds_stage | tile | walk ms | frame ms | covered pixels |", "|---|---|---|---|---|---|---|"]
for res, rows in (((2400, 1800), (838, 124)), ((2400, 1800), (0, 514)), ((2400, 1800), (776, 248)), ((4800, 3600), (1676, 248))):
    ctx.set_row_range(0, -1)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_row_range(*rows)
    for lds, tile in ((2, 3), (1, 3), (0, 3), (0, 0), (0, 1), (2, 0), (2, 1)):
        ctx.set_option("lds_stage", lds)
        ctx.set_option("tile", tile)
        st = timed()
        print(f"{res[0]}x{res[1]} rows {rows}: lds_stage {lds} tile {tile}: walk {st['ms_walk']:.4f} ms, frame {st['ms_total']:.4f}, "
              f"covered {st['covered_pixels']}", flush=True)
        lines.append(f"| {res[0]}x{res[1]} | {rows} | {lds} | {tile} | {st['ms_walk']:.4f} | {st['ms_total']:.4f} | {st['covered_pixels']} |")

from course5_amd.build import kernel_source_hash
lines.insert(2, f"kernel sources {kernel_source_hash()}")
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_share_probe.md"), "w").write("\n".join(lines) + "\n")
