"""The loop body of the product's walk kernel from its assembly, instructions per phase of the phase clock and by class, each
class weighted by its measured issue cost.

    python scripts/walk_isa.py [tag] [rates.json]      -> profiles/<tag>_walk_isa.md  (+ the loop's listing)

Runs here (hipcc cross-compiles): the kernel is compiled twice, as the product builds it and with -DC5_ISA_MARKERS=1, which
turns the phase clock's stamp points into assembly COMMENTS; the two loops must hold the same instructions.  Issue costs:
scripts/probes/valu_rate_probe.hip on the GPU (rates.json = its output as {name: cycles per wave-instruction and SIMD});
without the file the round-3 measurements are used.  Phases (walk_kernels.hip: C5_STAMP):
  0 election (lanes -> slots)   1 slot ids read back, LDS-DMA issued   2 (emission of the step before: empty in this kernel)
  3 waiting for the staging loads   4 record read, geometry, emission, exit face, (re-entry), loop control
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN2c518walk_composite_ldsILi3ELi0ELb1ELi14ELb1ELb0EEEvNS_10WalkParamsE"
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
# cycles a SIMD is held per wave-instruction (scripts/probes/valu_rate_probe.hip, 8 wavefronts per SIMD, independent chains)
RATES = {"valu64": 4.0, "valu32": 2.0, "valu_trans": 8.0, "salu": 1.0, "lds": 1.0, "vmem": 1.0, "branch": 1.0, "wait": 0.0, "other": 1.0}
if len(sys.argv) > 2:
    RATES.update(json.load(open(sys.argv[2])))


def compile_s(extra):
    out = f"/tmp/walk_isa_{len(extra)}.s"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "course5_amd", "csrc", "walk_kernels.hip"), "-o", out] + extra, check=True, capture_output=True)
    lines, on = [], False
    for ln in open(out):
        if ln.startswith(KERNEL + ":"):
            on = True
        if on:
            lines.append(ln.rstrip("\n"))
            if ln.startswith(".Lfunc_end"):
                break
    return lines


def classify(op):
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if re.search(r"_f64|_i64|_u64|_b64|lshl_add_u64|mad_i64|mad_u64", op):
            return "valu64"
        if re.search(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
            return "valu_trans"
        return "valu32"
    return "other"


def loop_of(lines):
    """(start, end) of the main loop: the Depth=1 loop that holds the LDS-DMA loads."""
    heads = [i for i, ln in enumerate(lines) if "Loop Header: Depth=1" in ln]
    for h in heads:
        label = lines[h].split(":")[0]
        num = label.split("_")[-1]
        # the loop ends at the last line that says "in Loop: Header=BB.._<num>"
        body = [i for i, ln in enumerate(lines) if f"Header=BB{label[4:].split('_')[0]}_{num} " in ln or ln.startswith(label + ":")]
        end = max(body)
        # extend to the end of that last block
        while end + 1 < len(lines) and not lines[end + 1].startswith(".LBB"):
            end += 1
        if any("global_load_lds_dwordx4" in lines[i] for i in range(h, end + 1)):
            return h, end
    raise SystemExit("main loop not found")


def instructions(lines, a, b):
    out = []
    for i in range(a, b + 1):
        ln = lines[i].strip()
        if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":") or re.match(r"^\.?L?BB", ln):
            if "C5_PHASE_END" in ln:
                out.append(("MARK", int(ln.split("C5_PHASE_END")[1].split()[0]), i))
            continue
        if ln.startswith(";;#"):
            continue
        op = ln.split()[0]
        out.append((op, ln, i))
    return out


plain = compile_s([])
marked = compile_s(["-DC5_ISA_MARKERS=1"])
pa, pb = loop_of(plain)
ma, mb = loop_of(marked)
ins_plain = [x for x in instructions(plain, pa, pb)]
ins_marked = instructions(marked, ma, mb)
ops_plain = sorted(x[0] for x in ins_plain)
ops_marked = sorted(x[0] for x in ins_marked if x[0] != "MARK")
same = ops_plain == ops_marked

# rare blocks of the marked listing: second election round, direct-load fall-back, (re-)entry — found by what they hold
def block_ranges(lines, a, b):
    blocks, cur = [], a
    for i in range(a, b + 1):
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", lines[i].strip()) and i > cur:
            blocks.append((cur, i - 1))
            cur = i
    blocks.append((cur, b))
    return blocks


rare_lines = set()
labels = {marked[i].split(":")[0]: i for i in range(ma, mb + 1) if re.match(r"^\.LBB\d+_\d+:", marked[i])}
RARE = ("offset:1024", "global_load_dwordx4 v[", "global_load_dwordx2 v[", "global_load_dwordx3", "v_ldexp_f64")
for i in range(ma, mb + 1):
    m = re.match(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", marked[i])
    if not m or m.group(1) not in labels or labels[m.group(1)] <= i:
        continue
    j = labels[m.group(1)]
    text = "\n".join(marked[i + 1:j])
    if "global_load_lds_dwordx4" in text or "v_fmac_f64" in text:  # (a region that holds the staging loads or the three planes IS the main path)
        continue
    # (a skipped region that holds the per-lane fall-back loads is the arm for "more distinct cells than slots", its own
    # copy of the eight LDS reads included: the main path has another copy behind the wave-uniform test)
    if "global_load_dwordx4 v[" in text or (any(r in text for r in RARE) and "ds_read_b128" not in text):
        rare_lines.update(range(i + 1, j))
# (the per-lane fall-back sits in the 'then' arm of a branch whose 'else' arm holds the eight ds_read_b128: only that arm is rare)
for (x, y) in block_ranges(marked, ma, mb):
    text = "\n".join(marked[x:y + 1])
    if "global_load_dwordx4 v[" in text and "ds_read_b128" not in text:
        rare_lines.update(range(x, y + 1))

phase = 0
table = {}   # phase -> class -> count (main path) ; rare counted apart
rare = {}
for op, ln, i in ins_marked:
    if op == "MARK":
        phase = ln + 1 if ln < 4 else 0  # what follows stamp 4 (loop control) is charged to phase 4's successor: the next step's phase 0
        continue
    cls = classify(op)
    tgt = rare if i in rare_lines else table
    tgt.setdefault(phase, {}).setdefault(cls, 0)
    tgt[phase][cls] += 1

names = {0: "0 election (tickets, winners, second table when buckets collide: apart)", 1: "1 leaders post ids, LDS-DMA issued", 2: "2 (empty)",
         3: "3 wait for the staging loads (s_waitcnt vmcnt(0))", 4: "4 record read, 3 planes, exit face, chord, tau, emission, next cell"}
classes = ["valu64", "valu32", "valu_trans", "salu", "lds", "vmem", "branch", "wait"]
md = [f"# {tag}: the loop of `walk_composite_lds<3, 0, true, 14, true>` (the headline's kernel) by phase and instruction class", "",
      f"`python scripts/walk_isa.py {tag}`; kernel sources {subprocess.run([sys.executable, '-c', 'from course5_amd.build import kernel_source_hash as h; print(h())'], cwd=ROOT, capture_output=True, text=True).stdout.strip()}; "
      f"the marked build's loop holds {'the SAME multiset of instructions as' if same else 'DIFFERENT instructions from'} the product's "
      f"({len(ops_plain)} against {len(ops_marked)}).", "",
      "Main path of one wavefront-step (every ray in a staged cell, no bucket collision, nobody leaves the grid); the rare blocks — second election "
      "table, per-lane loads of a cell beyond the 14 slots, (re-)entry through the entry lists — are counted apart below.", "",
      "| phase | " + " | ".join(classes) + " | all | issue cycles (weighted) |", "|---|" + "---|" * (len(classes) + 2)]
tot = {c: 0 for c in classes}
tot_cyc = 0.0
for ph in sorted(table):
    row = table[ph]
    cyc = sum(RATES.get(c, 1.0) * n for c, n in row.items())
    tot_cyc += cyc
    for c in classes:
        tot[c] += row.get(c, 0)
    md.append(f"| {names.get(ph, ph)} | " + " | ".join(str(row.get(c, 0)) for c in classes) + f" | {sum(row.values())} | {cyc:.0f} |")
md.append("| **sum** | " + " | ".join(str(tot[c]) for c in classes) + f" | {sum(tot.values())} | {tot_cyc:.0f} |")
md += ["", "rare blocks (instructions, not on the main path): " + ", ".join(f"phase {ph}: {sum(r.values())}" for ph, r in sorted(rare.items())), "",
       "Issue cost per wave-instruction and SIMD, cycles: " + ", ".join(f"{k} {v:g}" for k, v in RATES.items()) +
       " (`scripts/probes/valu_rate_probe.hip`: fp64 FMA / add / min / compare 4, 32-bit vector 2 at 8 wavefronts per SIMD; scalar, LDS and "
       "memory instructions are issued by their own ports, 1 cycle of the wavefront's issue slot each).", ""]
open(os.path.join(ROOT, "profiles", f"{tag}_walk_isa_counts.json"), "w").write(json.dumps({"main": table, "rare": rare, "rates": RATES, "same_as_product": same}, indent=1))
listing = ["", "## The loop (marked build; phase boundaries as comments)", "", "```"] + [marked[i] for i in range(ma, mb + 1)] + ["```"]
open(os.path.join(ROOT, "profiles", f"{tag}_walk_isa.md"), "w").write("\n".join(md + listing) + "\n")
print("\n".join(md))
