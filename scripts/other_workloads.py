"""Turn the bench lines of the secondary workloads (gpurun_out/bench_<tag>.json, written by the commands
listed below) into profiles/<round>_other_workloads.md.

    python bench.py                                                  > gpurun_out/bench_final.json
    python bench.py --lds-stage 0 --no-cpu-baseline                  > gpurun_out/bench_direct.json
    python bench.py --no-cpu-baseline --steps 360 --solids --sweep D > gpurun_out/bench_c5D.json   (and --sweep Y)
    python bench.py --no-cpu-baseline --res RES --steps 100          > gpurun_out/bench_RES.json
    python bench.py --no-cpu-baseline --workload c2 --res 1200x900   > gpurun_out/bench_c2.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
rows = [("C3 998 250-cell grid, 2400x1800 (the bench line)", "bench_final"),
        ("same, direct-load kernel (--lds-stage 0)", "bench_direct"),
        ("C3 grid, 1200x900", "bench_1200x900"),
        ("C3 grid, 4800x3600 (BASELINE config 4 on one GPU)", "bench_4800x3600"),
        ("C3 grid, 6788x5091 (8 GPUs' worth of rays on one)", "bench_6788x5091"),
        ("C2 ball (107k cells, non-convex), 1200x900", "bench_c2"),
        ("config 5 on one GPU: 360 frames, lobe + sphere, -D sweep", "bench_c5D"),
        ("config 5 on one GPU: 360 frames, lobe + sphere, -Y sweep", "bench_c5Y")]
out = [f"# {tag}: other workloads on one MI355X (python bench.py --no-cpu-baseline ...; same build as {tag}_bench.json)", "",
       "| workload | Mrays/s | ms per frame | walk kernel ms | segments per frame |", "|---|---|---|---|---|"]
for name, f in rows:
    path = os.path.join(ROOT, "gpurun_out", f + ".json")
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    out.append(f"| {name} | {d['value']:.0f} | {d['ms_per_step']:.4f} | {d['roofline']['kernel_ms']:.4f} | {d['config']['segments_per_frame']} |")
open(os.path.join(ROOT, "profiles", f"{tag}_other_workloads.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
