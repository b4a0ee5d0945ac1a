"""What row-splitting a frame over N GPUs can give at best, measured on ONE GPU: every rank's share of the frame
(cost-balanced contiguous blocks, and cyclic 16-row tiles) is rendered in turn and timed; the slowest share is the
frame time of an N-GPU run without its exchange.  A ray is a chain of ~130 dependent steps, so a share's time does not
fall in proportion to its rows.

    python scripts/sim_scaling.py [round-tag]      -> profiles/sim_scaling.json (read by bench.py: `predicted`)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402,F401  (HIP runtime load order)
from course5_amd import capi, meshgen as mg, sharding  # noqa: E402
from course5_amd.build import kernel_source_hash  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
base = float(os.environ.get("C5_ROW_BASE_COST", "3.0"))  # bench.py --row-base-cost
ctx = capi.Context(0)
out_dev = torch.zeros((3600, 4800, 2), dtype=torch.float32, device="cuda:0")  # room for the largest frame
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_view(mg.view_rotations(0.1, 0.07))
out = {"round": tag, "source_hash": kernel_source_hash(), "workload": "c3, view -X 0.1 -Y 0.07, fp64 walk", "row_base_cost": base,
       "what": "ms = GPU frame time (transform + records + entries + walk, HIP events) of the best of 8 renders after 30 warm ones; "
               "wall_ms = wall clock per frame over 200 frames back to back into device memory without stage events (how bench.py times a frame); "
               "option depth_split 0 (the default): small shares cut their rays in slabs by themselves",
       "frames": {}}


def timed():
    for _ in range(30):
        ctx.render()
    best = None
    for _ in range(8):
        ctx.render()
        st = ctx.stats()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    # ... and as bench.py times a frame: 200 frames back to back into device memory, no stage events, wall clock
    # (a frame's launches overlap the frame before on the host side; the per-stage events above cost a few per cent)
    ctx.set_option("stage_timing", 0)
    for _ in range(20):
        ctx.render_device(out_dev.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.render_device(out_dev.data_ptr())
    rc = ctx.synchronize()
    best["wall_ms"] = (time.perf_counter() - t0) * 1e3 / 200 if rc == 0 else float("nan")
    ctx.set_option("stage_timing", 1)
    return best


for res in ((2400, 1800), (4800, 3600)):
    ctx.set_row_tiles(0, 0, 1)
    ctx.set_row_range(0, -1)
    ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
    ctx.set_option("row_costs", 1)
    full = timed()
    costs = ctx.row_costs()
    ctx.set_option("row_costs", 0)
    full = timed()
    entry = {"full_ms": round(full["ms_total"], 4), "full_walk_ms": round(full["ms_walk"], 4), "full_wall_ms": round(full["wall_ms"], 4), "world": {}}
    print(f"{res[0]}x{res[1]}: full frame {full['ms_total']:.3f} ms (walk {full['ms_walk']:.3f})", flush=True)
    for world in (2, 4, 8):
        blocks = sharding.balanced_blocks(costs, world, base_cost=res[0] * base, quantum=8)
        per = []
        for b, n in blocks:
            ctx.set_row_range(b, n)
            per.append(timed())
        # ... and cut again by the TIMES just measured (what bench.py --gpus N and `course --devices` do with the times of a
        # probe frame: sharding.time_weighted_costs), twice; the best layout is kept
        for _ in range(2):
            again = sharding.balanced_blocks(sharding.time_weighted_costs(costs, blocks, [p["ms_total"] for p in per], base_cost=res[0] * base), world, quantum=8)
            if again == blocks:
                break
            per2 = []
            for b, n in again:
                ctx.set_row_range(b, n)
                per2.append(timed())
            if max(p["ms_total"] for p in per2) < max(p["ms_total"] for p in per):
                blocks, per = again, per2
        ctx.set_row_range(0, -1)
        tot = [round(p["ms_total"], 4) for p in per]
        wall = [round(p["wall_ms"], 4) for p in per]
        e = {"blocks": {"rows": [n for _, n in blocks], "ms": tot, "max_ms": max(tot), "speedup": round(full["ms_total"] / max(tot), 3),
                        "wall_ms": wall, "wall_max_ms": max(wall), "wall_speedup": round(full["wall_ms"] / max(wall), 3),
                        "steps": [p["steps"] for p in per],
                        "setup_ms": [round(p["ms_transform"] + p["ms_records"] + p["ms_entries"], 4) for p in per],
                        "walk_ms": [round(p["ms_walk"], 4) for p in per]}}
        per = []
        for r in range(world):
            ctx.set_row_tiles(16, r, world)
            per.append(timed())
        ctx.set_row_tiles(0, 0, 1)
        tot = [round(p["ms_total"], 4) for p in per]
        e["cyclic"] = {"ms": tot, "max_ms": max(tot), "speedup": round(full["ms_total"] / max(tot), 3)}
        entry["world"][str(world)] = e
        print(f"  N = {world}: blocks {e['blocks']['rows']} -> {e['blocks']['ms']} ms, x{e['blocks']['speedup']} (wall {e['blocks']['wall_ms']}, x{e['blocks']['wall_speedup']});  "
              f"cyclic -> max {e['cyclic']['max_ms']} ms, x{e['cyclic']['speedup']}", flush=True)
    out["frames"][f"{res[0]}x{res[1]}"] = entry
with open(os.path.join(ROOT, "profiles", "sim_scaling.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote profiles/sim_scaling.json")
