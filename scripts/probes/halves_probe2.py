"""Scratch probe: who ends last in the top half of the C3 frame (diagnostic build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from course5_amd import capi, meshgen as mg
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
lib = capi.load_library()
n_blocks = 131072
buf = (C.c_ulonglong * (4 * n_blocks))()
ctx.set_view(mg.view_rotations(0.1, 0.07))
S = 4; tiles_x = 300; sbx_n = 75
def tile_of(bid, sby_n):
    xcd = bid & 7; seq = bid >> 3
    sb = (seq // 16) * 8 + xcd; within = seq % 16
    sby, sbx = divmod(sb, sbx_n)
    return sbx * S + within % S, sby, within // S   # tx, position of the row of super-blocks in the order, row within it
for rows in ((0, 900), (900, 900)):
    for co in (1, 0):
        ctx.set_row_range(0, -1); ctx.set_row_range(*rows)
        ctx.set_option("cost_order", co)
        for _ in range(60): ctx.render()
        lib.c5_debug_walk_trace(buf, n_blocks, 1)
        ctx.render(); st = ctx.stats()
        lib.c5_debug_walk_trace(buf, n_blocks, 1)
        t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
        idx = np.nonzero((t[:, 1] > 0) & (t[:, 3] > 0))[0]
        b = t[idx, 0].astype(np.int64); e = t[idx, 1].astype(np.int64); steps = t[idx, 3].astype(np.int64)
        t0 = b.min(); b = (b - t0) / 100.0; e = (e - t0) / 100.0
        print(f"rows {rows} cost_order {co}: walk {st['ms_walk']:.4f}; jobs {len(idx)}; last end {e.max():.1f}")
        last = np.argsort(-e)[:14]
        for k in last:
            tx, pos, r = tile_of(int(idx[k]), 29)
            print(f"     block {idx[k]:6d} (tx {tx:3d}, super-block row #{pos:2d} in the order, tile row {r}) start {b[k]:6.1f} end {e[k]:6.1f} steps {steps[k]:4d} ns/step {(e[k]-b[k])*1e3/steps[k]:5.0f}")
        # per position in the order: steps median and start
        pos = np.array([tile_of(int(i), 29)[1] for i in idx])
        print("     row of super-blocks #: jobs, median steps, median start, max end")
        for p in range(0, 29):
            m = pos == p
            if m.any(): print(f"       #{p:2d}: {m.sum():4d} jobs, steps {np.median(steps[m]):5.0f}, start {np.median(b[m]):6.1f}, end max {e[m].max():6.1f}, ns/step {np.median((e[m]-b[m])*1e3/steps[m]):5.0f}")
ctx.set_option("cost_order", 1)
