"""Turn the bench lines of the secondary workloads (gpurun_out/other/*.json, written by scripts/other_workloads.sh on
the GPU box) into profiles/<round>_other_workloads.md; the first row is the round's own bench line
(profiles/<round>_bench.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = [("C3 998 250-cell grid, 2400x1800 (the bench line)", os.path.join(ROOT, "profiles", f"{tag}_bench.json")),
        ("same, staging through vector registers (--lds-stage 1)", "stage1"),
        ("same, direct per-lane loads (--lds-stage 0)", "stage0"),
        ("C3 grid, 1200x900", "c3_1200x900"),
        ("C3 grid, 4800x3600 (BASELINE config 4 on one GPU)", "c3_4800x3600"),
        ("C2 ball (107k cells, non-convex), 1200x900", "c2_1200x900"),
        ("config 5 on one GPU: 360 frames, lobe + sphere, -D sweep", "c5_D"),
        ("config 5 on one GPU: 360 frames, lobe + sphere, -Y sweep", "c5_Y")]
out = [f"# {tag}: other workloads on one MI355X (scripts/other_workloads.sh: python bench.py --no-cpu-baseline --no-native --steps 200 ...; "
       f"same build as {tag}_bench.json)", "",
       "| workload | Mrays/s | ms per frame | walk kernel ms | delivered to host: Mrays/s | segments per frame |",
       "|---|---|---|---|---|---|"]


def last_json_line(path):
    lines = [ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


for name, f in rows:
    path = f if os.path.isabs(f) else os.path.join(ROOT, "gpurun_out", "other", f + ".json")
    if not os.path.exists(path):
        continue
    d = last_json_line(path)
    h = (d.get("value_host_image") or {}).get("pipelined") or {}
    fmt = lambda v, spec: format(v, spec) if v is not None else "-"  # noqa: E731
    out.append(f"| {name} | {d['value']:.0f} | {d['ms_per_step']:.4f} | {d['roofline']['kernel_ms']:.4f} | "
               f"{fmt(h.get('value'), '.0f')} | {d['config']['segments_per_frame']} |")
open(os.path.join(ROOT, "profiles", f"{tag}_other_workloads.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
