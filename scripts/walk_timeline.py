"""Launch timeline of walk_composite_lds on the C3 frame (library built with -DC5_WALK_STAMPS=1): when every wavefront
with rays started and ended (s_memrealtime, 100 MHz), on which XCD; from it the wavefront slots in use over the launch,
per-XCD finish times and how much of the launch is ramp and tail.

    scripts/build_variant.sh stamps -DC5_WALK_STAMPS=1
    C5_LIB=course5_amd/libcourse5_hip_stamps.so python scripts/walk_timeline.py [round-tag]  -> profiles/<tag>_walk_timeline.md
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg  # noqa: E402
from course5_amd.build import kernel_source_hash  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
opts = dict(kv.split("=") for kv in os.environ.get("C5_OPTS", "").split(",") if kv)
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
res = tuple(int(v) for v in os.environ.get("C5_RES", "2400x1800").split("x"))
ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
for k, v in opts.items():
    ctx.set_option(k, float(v))
lib = capi.load_library()
n_blocks = 131072
SLOTS = 256 * 4 * int(os.environ.get("C5_WAVES", "8"))  # CUs x SIMDs x resident wavefronts per SIMD (launch bounds of the default kernel)
buf = (C.c_ulonglong * (4 * n_blocks))()
out = torch.zeros((res[1], res[0], 2), dtype=torch.float32, device="cuda:0")
for _ in range(300):
    ctx.render_device(out.data_ptr())
ctx.synchronize()
lib.c5_debug_walk_trace(buf, n_blocks, 1)
ctx.set_option("stage_timing", 1)
ctx.render()
st = ctx.stats()
lib.c5_debug_walk_trace(buf, n_blocks, 1)
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
t = t[t[:, 1] > 0]
b, e = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
xcc = (t[:, 2] & 0xff).astype(int)
steps = t[:, 3].astype(np.int64)
t0 = b.min()
b, e = (b - t0) / 100.0, (e - t0) / 100.0  # microseconds
span = e.max()
lines = [f"# {tag}: launch timeline of the walk on the C3 frame ({res[0]}x{res[1]}), options {opts or 'default'}", "",
         f"kernel sources {kernel_source_hash()}; walk {st['ms_walk']:.3f} ms by HIP events; {len(t)} wavefronts with rays, first start to last end {span:.1f} us; "
         f"wavefront lifetime: median {np.median(e - b):.1f} us, 90 % {np.percentile(e - b, 90):.1f}, max {(e - b).max():.1f}; "
         f"steps per wavefront: median {np.median(steps):.0f}, max {steps.max()}; {np.median((e - b) * 1e3 / np.maximum(steps, 1)):.0f} ns per wavefront-step", ""]
# slots in use over time
grid = np.linspace(0, span, 41)
lines += [f"| time (us) | wavefronts with rays resident | of {SLOTS} slots |", "|---|---|---|"]
for x in grid[:-1] + (grid[1] - grid[0]) / 2:
    n = int(((b <= x) & (e > x)).sum())
    lines.append(f"| {x:.0f} | {n} | {n / SLOTS:.2f} |")
busy = (e - b).sum()
lines += ["", f"slot-time used by wavefronts with rays: {busy:.0f} us = {busy / (SLOTS * span):.3f} of {SLOTS} slots x {span:.1f} us", ""]
lines += ["| XCD | wavefronts | wavefront-steps | first start (us) | last end (us) | slot-time (us) |", "|---|---|---|---|---|---|"]
for x in sorted(set(xcc)):
    m = xcc == x
    lines.append(f"| {x} | {int(m.sum())} | {int(steps[m].sum())} | {b[m].min():.1f} | {e[m].max():.1f} | {(e[m] - b[m]).sum():.0f} |")
order = np.sort(e)
# throughput over the launch: a wavefront's steps spread evenly over its life
edges = np.linspace(0, span, 41)
rate = np.zeros(40)
for k in range(40):
    lo, hi = edges[k], edges[k + 1]
    overlap = np.clip(np.minimum(e, hi) - np.maximum(b, lo), 0, None)
    rate[k] = (steps * overlap / np.maximum(e - b, 1e-9)).sum() / (hi - lo)
steady = rate[2:int(0.7 * 40)].mean()
lines += ["", f"wavefront-steps per us: steady state (5 % - 70 % of the launch) {steady:.0f}, whole launch {steps.sum() / span:.0f} "
              f"(ratio {steady * span / steps.sum():.3f}); last fifth of the launch {rate[32:].mean():.0f}", ""]
summary = {"round": tag, "source_hash": kernel_source_hash(), "span_us": round(float(span), 1), "wavefronts": int(len(t)),
           "wave_steps": int(steps.sum()), "slot_time_frac": round(float(busy / (SLOTS * span)), 4),
           "steady_steps_per_us": round(float(steady), 1), "mean_steps_per_us": round(float(steps.sum() / span), 1),
           "steady_over_mean": round(float(steady * span / steps.sum()), 4),
           "resident_frac_steady": round(float(np.mean([((b <= x) & (e > x)).sum() for x in edges[2:28]]) / SLOTS), 4),
           "tail_starts_us": round(float(order[int(len(order) * 0.5)]), 1) if False else None,
           "median_wave_life_us": round(float(np.median(e - b)), 1), "source": f"profiles/{tag}_walk_timeline.md"}
# tail: when does the number of resident wavefronts fall below half the slots for good
lines += ["", f"ends: 50 % of the wavefronts have ended by {order[len(order) // 2]:.1f} us, 90 % by {order[int(len(order) * 0.9)]:.1f}, "
              f"99 % by {order[int(len(order) * 0.99)]:.1f}, all by {span:.1f}",
          f"starts: last wavefront with rays starts at {b.max():.1f} us"]
text = "\n".join(lines) + "\n"
print(text)
if not opts and res == (2400, 1800):
    with open(os.path.join(ROOT, "profiles", f"{tag}_walk_timeline.md"), "w") as f:
        f.write(text)
    summary.pop("tail_starts_us")
    with open(os.path.join(ROOT, "profiles", f"{tag}_walk_timeline.json"), "w") as f:
        json.dump(summary, f, indent=1)
