"""In-kernel phase clock of walk_composite_lds (library built with -DC5_WALK_STAMPS=1, see walk_kernels.hip):

    scripts/build_variant.sh stamps -DC5_WALK_STAMPS=1
    C5_LIB=course5_amd/libcourse5_hip_stamps.so python scripts/stamp_walk.py [round-tag]   -> profiles/<tag>_walk_phases.md

One wavefront in 67 (a stride that visits every XCD) sums, per phase of a step, the s_memtime ticks (= shader cycles, MI355X_MICROARCH.md) between
stamps; it drains vmcnt / lgkmcnt at the two stamps that end a phase of waiting, so its split is exact; the other 63
run the product's instruction stream, so the sampled wavefronts see the machine as loaded as the product does.
"""
import ctypes as C
import json
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import capi, meshgen as mg  # noqa: E402
from course5_amd.build import kernel_source_hash  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(0.1, 0.07))
rows = os.environ.get("C5_ROWS")  # "first,count": the phase clock of a small share of the frame (few wavefronts per SIMD); nothing written
if rows:
    ctx.set_row_range(*(int(v) for v in rows.split(",")))
lib = capi.load_library()
if not hasattr(lib, "c5_debug_walk_stamps"):
    raise SystemExit("this library was not built with -DC5_WALK_STAMPS=1 (scripts/build_variant.sh stamps -DC5_WALK_STAMPS=1)")
out = torch.zeros((ctx.local_rows if rows else 1800, 2400, 2), dtype=torch.float32, device="cuda:0")
buf = (C.c_ulonglong * 16)()
names = ["election: ticket -> leader -> slot (2 LDS round trips)",
         "slot ids read back + staging loads (LDS-DMA) issued",
         "(emission of the previous step in the loads' shadow: only in builds with C5_EMIT_NOW=0)",
         "staging loads land: s_waitcnt vmcnt(0)",
         "8 x ds_read_b128 + three exit planes + exit face (+ re-entry) + this step's emission / absorption (exp)"]
short = ["election", "issue", "deferred emission", "load wait", "read + geometry + emission"]
lines = [f"# {tag}: phase clock of walk_composite_lds<3, 0, true, 14, true> (exit records, short exp series only, 8 wavefronts per SIMD) on the C3 frame (2400x1800, fp64 walk)", "",
         "`scripts/build_variant.sh stamps -DC5_WALK_STAMPS=1 && C5_LIB=course5_amd/libcourse5_hip_stamps.so python scripts/stamp_walk.py`",
         f"kernel sources: {kernel_source_hash()}", ""]
result = {}
for stage, label in ((2, "LDS-DMA staging (default)"), (1, "staged through vector registers")):
    ctx.set_option("tile", 3)
    ctx.set_option("lds_stage", stage)
    for _ in range(300):  # sustained clocks (the first frames after an idle spell run slower)
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    lib.c5_debug_walk_stamps(buf, 1)
    ctx.render()
    st = ctx.stats()
    lib.c5_debug_walk_stamps(buf, 1)
    v = list(buf)
    tot = sum(v[:5])
    steps = max(v[12], 1)
    lines += [f"## lds_stage {stage}: {label}", "",
              f"walk {st['ms_walk']:.3f} ms (stamped build: one wavefront in 67 stamps), {st['steps']} lane-steps in the frame; sampled: "
              f"{v[9]} wavefronts with rays, {v[12]} wavefront-steps ({v[13] / steps:.1f} walking lanes per step), loop "
              f"{v[8] / max(v[9], 1):.0f} cycles per wavefront, **{tot / steps:.0f} shader cycles per wavefront-step**", "",
              "| phase | cycles | share | cycles per wavefront-step |", "|---|---|---|---|"]
    for n, x in zip(names, v[:5]):
        lines.append(f"| {n} | {x} | {100.0 * x / tot:.1f} % | {x / steps:.2f} |")
    if v[15]:
        lines.append(f"\n(-DC5_WALK_STAMPS=2) distinct cells per wavefront-step {v[11] / steps:.2f} (runs of equal ids {v[10] / steps:.2f}); "
                     f"a ray's NEXT cell is one this step already staged for some lane: {100.0 * v[14] / v[15]:.1f} % of {v[15]} lane-steps")
    lines.append("")
    if stage == 2:
        k = max(range(5), key=lambda i: v[i])
        result = {"cycles_per_wave_step": round(tot / steps, 1), "cycles": {short[i]: round(v[i] / steps, 1) for i in range(5)},
                  "share": {short[i]: round(v[i] / tot, 4) for i in range(5)}, "dominant": short[k],
                  "walking_lanes_per_step": round(v[13] / steps, 2), "source": f"profiles/{tag}_walk_phases.md"}
text = "\n".join(lines) + "\n"
print(text)
if rows:
    raise SystemExit(0)
with open(os.path.join(ROOT, "profiles", f"{tag}_walk_phases.md"), "w") as f:
    f.write(text)
with open(os.path.join(ROOT, "profiles", f"{tag}_walk_phases.json"), "w") as f:
    json.dump(result, f, indent=1)
