#!/bin/bash
# round 4, final GPU run: the whole GPU suite on the final tree, smoke, then the committed profile set (scripts/final_profile.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" > gpurun_out/final/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final/summary.txt
bash scripts/final_profile.sh r04 > gpurun_out/final/final_profile.log 2>&1
cat gpurun_out/final/summary.txt; tail -3 gpurun_out/final/pytest_gpu.log; tail -3 gpurun_out/final/smoke.log; tail -c 700 gpurun_out/final/bench.json
