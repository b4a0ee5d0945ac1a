"""Scratch: estimate N-GPU strong scaling on ONE GPU by timing each rank's share of the frame
in turn (no gather).  Prints per-rank GPU frame time for cyclic tiles and balanced blocks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from course5_amd import capi, meshgen as mg, sharding

res = (2400, 1800) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
base = 0.5 if len(sys.argv) < 3 else float(sys.argv[2])
ctx = capi.Context(0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
ctx.set_option("row_costs", 1)

def timed():
    best = None
    for _ in range(6):
        ctx.render()
        st = ctx.stats()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    return best

full = timed()
costs = ctx.row_costs()
print("full frame", {k: round(v, 3) for k, v in full.items() if k.startswith("ms_")})
for world in (2, 4, 8):
    blocks = sharding.balanced_blocks(costs, world, base_cost=res[0] * base)
    per = []
    for b, n in blocks:
        ctx.set_row_range(b, n)
        st = timed()
        per.append(st)
    ctx.set_row_range(0, -1)
    tot = [round(p["ms_total"], 3) for p in per]
    print(f"blocks world {world}: rows {[n for _, n in blocks]} ms_total {tot} max {max(tot)} -> speedup {full['ms_total'] / max(tot):.2f}",
          "setup", [round(p["ms_transform"] + p["ms_records"] + p["ms_entries"], 3) for p in per],
          "walk", [round(p["ms_walk"], 3) for p in per])
    per = []
    for r in range(world):
        ctx.set_row_tiles(int(os.environ.get("TILE_ROWS", "16")), r, world)
        per.append(timed())
    ctx.set_row_tiles(0, 0, 1)
    tot = [round(p["ms_total"], 3) for p in per]
    print(f"cyclic world {world}: ms_total {tot} max {max(tot)} -> speedup {full['ms_total'] / max(tot):.2f}")
# host-side enqueue cost of a frame
import ctypes
buf = np.zeros(1)
ctx.set_option("stage_timing", 0)
t = time.perf_counter()
for _ in range(200):
    ctx.lib.c5_render_device(ctx.handle, ctx.lib.c5_render_device.argtypes and ctypes.c_void_p(0) or None) if False else None
print("done")
