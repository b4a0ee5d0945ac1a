#!/bin/bash
# Bench lines for the workloads other than the headline one (profiles/<round>_other_workloads.md is written from these).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
OUT=gpurun_out/other
mkdir -p "$OUT"
B="python3 bench.py --no-cpu-baseline --no-native --steps 200"
$B --lds-stage 1 --no-host-image > $OUT/stage1.json 2> $OUT/err.log || exit 1
$B --lds-stage 0 --no-host-image > $OUT/stage0.json 2>> $OUT/err.log || exit 1
$B --res 1200x900 --no-host-image > $OUT/c3_1200x900.json 2>> $OUT/err.log || exit 1
$B --res 4800x3600 > $OUT/c3_4800x3600.json 2>> $OUT/err.log || exit 1
$B --workload c2 --res 1200x900 > $OUT/c2_1200x900.json 2>> $OUT/err.log || exit 1
$B --solids --sweep D --steps 360 --no-host-image > $OUT/c5_D.json 2>> $OUT/err.log || exit 1
$B --solids --sweep Y --steps 360 --no-host-image > $OUT/c5_Y.json 2>> $OUT/err.log || exit 1
for f in $OUT/*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
h = (d.get("value_host_image") or {}).get("pipelined") or {}
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], "host", h.get("value"), h.get("ms_per_frame"))
PY
done
