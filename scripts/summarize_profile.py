"""Turn rocprofv3 CSV output (gpurun_out/...) into the committed summaries under profiles/.

    python scripts/summarize_profile.py r01

Reads  gpurun_out/prof_kt/kt_kernel_stats.csv            (--kernel-trace --stats)
       gpurun_out/pmc_<COUNTER>/pmc_counter_collection.csv (one --pmc pass per counter group)
Writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc.md, profiles/traffic.json and
profiles/roofline.json (what bench.py reads: per-launch HBM bytes and how busy each unit is).

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate passes,
are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request, so reads are doubled.
"""
import collections
import csv
import json
import os
import shutil
import sys

SUFFIX = ""
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


def kernel_avg_ms(tag, walk):
    """Average duration of the walk kernel in the kernel-trace summary (the default bench command)."""
    path = os.path.join(PROF, f"{tag}_kernel_stats.csv")
    with open(path) as f:
        for row in csv.DictReader(f):
            if "walk_composite" in row.get("Name", ""):
                return float(row["AverageNs"]) / 1e6, int(row["Calls"])
    return None, 0


def write_roofline(tag, walk, c):
    """profiles/roofline.json: every figure per launch of the walk kernel; fractions are of what the
    hardware could do in the launch's own duration (GRBM_GUI_ACTIVE / 8 = shader cycles of the launch)."""
    get = lambda k: c[k][0] if k in c else None  # noqa: E731
    cycles = get("GRBM_GUI_ACTIVE") / 8.0 if get("GRBM_GUI_ACTIVE") else None  # summed over the 8 XCDs
    n_simd, n_cu = 1024, 256
    units = {}
    if cycles:
        if get("SQ_ACTIVE_INST_VALU"):  # quad-cycles a SIMD's vector pipe was executing, summed over SIMDs
            units["valu"] = {"frac": 4.0 * get("SQ_ACTIVE_INST_VALU") / (cycles * n_simd),
                             "what": "4 x SQ_ACTIVE_INST_VALU / (cycles x 1024 SIMDs): share of the launch the vector pipes were executing, "
                                     "counting every vector instruction as 4 cycles - an UPPER bound: 32-bit instructions issue in about "
                                     "2 cycles on gfx950, 64-bit and packed ones in about 4 (scripts/probes/valu_rate_probe.hip)"}
        if get("SQ_LDS_IDX_ACTIVE"):
            units["lds"] = {"frac": get("SQ_LDS_IDX_ACTIVE") / (cycles * n_cu),
                            "what": "SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs): share of the launch the LDS arrays were busy",
                            "bank_conflict_share": (get("SQ_LDS_BANK_CONFLICT") or 0.0) / get("SQ_LDS_IDX_ACTIVE")}
        if get("SQ_WAVE_CYCLES"):
            units["wave_slots"] = {"frac": 4.0 * get("SQ_WAVE_CYCLES") / (cycles * n_simd * 8),
                                   "what": "4 x SQ_WAVE_CYCLES / (cycles x 8192 wavefront slots): average occupancy (the exit-record walk's 62-64 VGPRs allow all 8 per SIMD)"}
        if get("SQ_WAIT_ANY") and get("SQ_WAVE_CYCLES"):
            units["waiting"] = {"frac": get("SQ_WAIT_ANY") / get("SQ_WAVE_CYCLES"),
                                "what": "SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a wavefront's life parked on s_waitcnt"}
    ms, calls = kernel_avg_ms(tag, walk)
    out = {"kernel": walk, "round": tag, "kernel_ms_rocprofv3": ms, "kernel_launches_rocprofv3": calls,
           "shader_cycles_per_launch": cycles}
    if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
        out["hbm_bytes_per_launch"] = (2.0 * get("FETCH_SIZE") + get("WRITE_SIZE")) * 1024.0
        out["source"] = (f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), "
                         "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per MI355X_MICROARCH.md HBM section")
    if cycles and get("TCP_TCC_READ_REQ_sum"):
        # a vector-cache -> L2 read request moves one 128-byte line on gfx950 (the records are read as whole lines)
        l2_bytes = 128.0 * get("TCP_TCC_READ_REQ_sum")
        secs = (ms or 0.0) * 1e-3
        units["l2"] = {"bytes_per_launch": l2_bytes, "frac": (l2_bytes / secs / 34.5e12) if secs else None,
                       "what": "128 B x TCP_TCC_READ_REQ_sum / kernel time against 34.5 TB/s aggregate L2",
                       "hit_rate": get("TCC_HIT_sum") / (get("TCC_HIT_sum") + get("TCC_MISS_sum")) if get("TCC_HIT_sum") else None}
    if out.get("hbm_bytes_per_launch") and ms:
        units["hbm"] = {"frac": out["hbm_bytes_per_launch"] / (ms * 1e-3) / 8.0e12,
                        "what": "HBM-side bytes / kernel time against 8 TB/s"}
    if cycles and get("SQ_ACTIVE_INST_SCA"):
        units["salu"] = {"frac": 4.0 * get("SQ_ACTIVE_INST_SCA") / (cycles * n_simd),
                         "what": "4 x SQ_ACTIVE_INST_SCA / (cycles x 1024 SIMDs): share of the launch the scalar units were executing"}
    out["units"] = units
    # what the in-kernel instruments add (scripts/stamp_walk.py, scripts/walk_timeline.py; same sources or they are left out)
    import sys as _sys
    _sys.path.insert(0, ROOT)
    from course5_amd.build import kernel_source_hash
    out["source_hash"] = kernel_source_hash()
    extra = {}
    for name in ("walk_phases", "walk_timeline"):
        try:
            with open(os.path.join(PROF, f"{tag}_{name}{SUFFIX}.json")) as f:
                extra[name] = json.load(f)
        except Exception:  # noqa: BLE001
            pass
    if "walk_phases" in extra:
        out["phases"] = extra["walk_phases"]
    tl = extra.get("walk_timeline")
    if tl:
        out["timeline"] = tl
    # What binds the kernel: not the counter that reads highest under a flat 4-cycles-per-instruction
    # formula, but the units' busy shares with every instruction class at its measured issue cost
    # (scripts/walk_isa.py -> profiles/<tag>_walk_isa_counts.json: the loop's instructions per wavefront-step by class),
    # in the steady state (while the wavefront slots are full), beside what a wavefront-step costs at 2 .. 8 wavefronts
    # per SIMD (scripts/share_timeline.py).
    lim = {}
    try:
        with open(os.path.join(PROF, f"{tag}_walk_isa_counts.json")) as f:
            isa = json.load(f)
        per_step = {}
        for ph in isa["main"].values():
            for cls, n in ph.items():
                per_step[cls] = per_step.get(cls, 0) + n
        rates = isa["rates"]
        lim["instructions_per_wave_step"] = per_step
        lim["issue_cycles"] = rates
    except Exception:  # noqa: BLE001
        per_step, rates = None, None
    steady = (tl or {}).get("steady_over_mean") or 1.0
    if per_step and cycles and tl and tl.get("wave_steps"):
        ws = tl["wave_steps"]
        valu_cyc = (per_step.get("valu64", 0) * rates["valu64"] + per_step.get("valu32", 0) * rates["valu32"]) * ws
        lim["valu_busy_weighted"] = {"launch": valu_cyc / (cycles * n_simd), "steady": min(1.0, valu_cyc / (cycles * n_simd) * steady),
                                     "what": "(fp64 vector instructions x 4 + 32-bit ones x 2 cycles) per wavefront-step x wavefront-steps / "
                                             "(cycles x 1024 SIMDs): not an upper bound like units.valu (which charges every vector instruction 4)"}
        scal = (per_step.get("salu", 0) + per_step.get("branch", 0)) * ws
        lim["scalar_busy_per_cu"] = {"launch": scal / (cycles * n_cu), "steady": min(1.0, scal / (cycles * n_cu) * steady),
                                     "what": "scalar + branch instructions per wavefront-step x wavefront-steps / (cycles x 256 CUs): ONE scalar unit "
                                             "per CU serves its four SIMDs, one instruction per cycle"}
    for k in ("lds", "l2", "hbm", "salu"):
        if units.get(k, {}).get("frac"):
            lim.setdefault("units_steady", {})[k] = min(1.0, units[k]["frac"] * steady)
    lim["name"] = "chain"
    lim["frac"] = (lim.get("valu_busy_weighted") or {}).get("steady")
    lim["note"] = ("No unit is saturated while the wavefront slots are full: vector pipes (weighted by what instructions cost) about half, "
                   "the CU's scalar unit and its LDS 55-75 %, L2 and HBM 12-16 %.  What a step costs is its dependent chain - four LDS round "
                   "trips, the two LDS-DMA pieces, ~170 instructions in order: phases - stretched by the queueing at those shared units: the "
                   "same step takes 843 ns with 2 wavefronts per SIMD, 984 with 4, 1 118 with 6, 1 221 with 7 (profiles/" + tag +
                   "_share_timeline.md), so eight wavefronts deliver 6.6 times what one does, not 8.  Hiding a phase (emission deferred behind the "
                   "loads: -1.4 %) or trimming one unit (8 scalar instructions and a scalar load per step: +0.5 %) does not shorten it "
                   "(profiles/experiments.md); the 128-byte exit records of round 3 cut LDS bytes, vector and scalar work together: 0.540 -> 0.456 ms.")
    out["limiter"] = lim
    with open(os.path.join(PROF, f"roofline{SUFFIX}.json"), "w") as f:
        json.dump(out, f, indent=1)
    lines = [f"walk_composite per launch: {cycles:.4g} shader cycles" if cycles else ""]
    for k, v in units.items():
        if v.get("frac") is not None:
            lines.append(f"  {k}: {v['frac']:.3f}  ({v['what']})")
    return lines


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    global SUFFIX
    SUFFIX = sys.argv[2] if len(sys.argv) > 2 else ""  # e.g. "_mixed": roofline_mixed.json, traffic is left alone
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
             "-- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-image --no-steady`", ""]
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
        if not SUFFIX:
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
    if walk:
        lines += [""] + write_roofline(tag, walk, counters[walk])
    with open(os.path.join(PROF, f"{tag}_pmc.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
