"""Turn rocprofv3 CSV output (gpurun_out/...) into the committed summaries under profiles/.

    python scripts/summarize_profile.py r01

Reads  gpurun_out/prof_kt/kt_kernel_stats.csv            (--kernel-trace --stats)
       gpurun_out/pmc_<COUNTER>/pmc_counter_collection.csv (one --pmc pass per counter group)
Writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc.md, profiles/traffic.json.

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate passes,
are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request, so reads are doubled.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def per_kernel_counter(path):
    """{kernel short name: {counter: [values per dispatch]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "").replace("c5::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(PROF, exist_ok=True)
    shutil.copy(os.path.join(OUT, "prof_kt", "kt_kernel_stats.csv"), os.path.join(PROF, f"{tag}_kernel_stats.csv"))
    counters = collections.defaultdict(dict)
    for d in sorted(os.listdir(OUT)):
        p = os.path.join(OUT, d, "pmc_counter_collection.csv")
        if not (d.startswith("pmc_") and os.path.exists(p)):
            continue
        for kern, cs in per_kernel_counter(p).items():
            for c, vals in cs.items():
                counters[kern][c] = (sum(vals) / len(vals), len(vals))
    lines = [f"# {tag}: rocprofv3 PMC summary (mean per dispatch; separate passes per counter group)", "",
             "Command per pass: `rocprofv3 --pmc <counters> --kernel-trace -d gpurun_out/pmc_X -o pmc --output-format csv "
             "-- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline`", ""]
    names = sorted({c for k in counters.values() for c in k})
    lines.append("| kernel | " + " | ".join(names) + " |")
    lines.append("|---|" + "---|" * len(names))
    for kern in sorted(counters):
        if kern.startswith("__amd") or kern.startswith("at::"):
            continue
        lines.append(f"| {kern} | " + " | ".join(f"{counters[kern][c][0]:.6g}" if c in counters[kern] else "" for c in names) + " |")
    walk = next((k for k in counters if k.startswith("walk_composite")), None)
    if walk and "FETCH_SIZE" in counters[walk] and "WRITE_SIZE" in counters[walk]:
        fetch_kib, _ = counters[walk]["FETCH_SIZE"]
        write_kib, _ = counters[walk]["WRITE_SIZE"]
        hbm = (2.0 * fetch_kib + write_kib) * 1024.0
        traffic = {"kernel": walk, "hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": fetch_kib,
                   "write_size_kib": write_kib,
                   "source": f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), "
                             "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per MI355X_MICROARCH.md HBM section"}
        with open(os.path.join(PROF, "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
        lines += ["", f"walk_composite HBM bytes per launch = (2 x {fetch_kib:.0f} + {write_kib:.0f}) KiB = {hbm / 1e6:.1f} MB"]
        if "TCC_HIT_sum" in counters[walk]:
            h, m = counters[walk]["TCC_HIT_sum"][0], counters[walk]["TCC_MISS_sum"][0]
            lines.append(f"walk_composite L2 hit rate = {h / (h + m):.4f}")
        c = counters[walk]
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            lines.append(f"walk_composite VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) "
                         f"= {c['SQ_THREAD_CYCLES_VALU'][0] / (64 * c['SQ_ACTIVE_INST_VALU'][0]):.3f}")
    with open(os.path.join(PROF, f"{tag}_pmc.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
