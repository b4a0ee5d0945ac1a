#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   scripts/collect_pmc.sh            # kernel trace + every PMC group, one pass each
# Outputs land under gpurun_out/ (prof_kt/, pmc_<group>/); scripts/summarize_profile.py <tag> turns
# them into the committed files under profiles/.  Counters are collected WITHOUT any trace domain
# other than --kernel-trace, one group per pass (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"
export TMPDIR=/tmp
cd "$ROOT" || exit 1
EXTRA=()
CMD=(python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-image --no-steady --no-native "${EXTRA[@]}")
GROUPS_=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum"
  "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum"
)
rm -rf "$OUT"/pmc_* "$OUT"/prof_kt
# the kernel trace runs the SAME command as the bench line (default steps / warmup), so that its average
# kernel duration and bench.py's HIP-event figure describe the same sustained state
rocprofv3 --kernel-trace --stats -d "$OUT/prof_kt" -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-host-image --no-native "${EXTRA[@]}" > "$OUT/prof_kt.log" 2>&1 || exit 1
for g in "${GROUPS_[@]}"; do
  name="pmc_${g%% *}"
  # shellcheck disable=SC2086
  rocprofv3 --pmc $g --kernel-trace -d "$OUT/$name" -o pmc --output-format csv -- "${CMD[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  echo "done $name"
done
