#!/bin/bash
# The whole GPU suite + smoke + the default bench line on one box:  bash scripts/gpu_suite.sh <out-dir-under-gpurun_out>
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" > $out/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/summary.txt
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; tail -5 $out/pytest_gpu.log; tail -3 $out/smoke.log; tail -c 2500 $out/bench.json
