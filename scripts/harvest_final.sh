#!/bin/bash
# Copy the file set of scripts/r6_final.sh (r5_final.sh before it) from gpurun_out/final/ into profiles/ under the round's prefix.
#   bash scripts/harvest_final.sh r05      (run in the container after the gpurun call has merged its outputs back)
# Fails if the run was not green or the bench line has no roofline.traffic.
set -euo pipefail
cd "$(dirname "$0")/.."
R=${1:?round prefix, e.g. r05}
F=gpurun_out/final
grep -q "rc=[^0]" $F/summary.txt && { echo "final run not green:"; cat $F/summary.txt; exit 1; }
python3 - <<PY
import json, sys
r = json.loads(open("$F/bench.json").read().strip().splitlines()[-1])
if r["roofline"].get("traffic") is None:
    sys.exit("roofline.traffic is null in $F/bench.json")
PY
for f in $F/*.json $F/*.csv $F/*_table*.txt $F/*_ab.txt $F/host_time.txt $F/pmc_conv.txt $F/pmc_dominant.txt $F/trained_weights_parity.txt $F/x3_wgrad_stamps.txt; do
    [ -f "$f" ] || continue
    cp "$f" profiles/${R}_final_$(basename "$f")
done
{ cat $F/summary.txt; grep -n "passed\|failed" $F/pytest_gpu.log | tail -1; grep "^smoke" $F/smoke.log || cat $F/smoke.log | tail -5; } > profiles/${R}_final_gpu_suite_summary.txt
ls profiles | grep -c "^${R}_final_"
