#!/bin/bash
# round 5, final GPU run: the whole GPU suite on the final tree (+ the trained-weights parity test with its printed numbers), smoke,
# the committed profile set (scripts/final_profile.sh r05), the split-precision weight gradient's table and phase stamps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" > gpurun_out/final/summary.txt
timeout 1500 python -m pytest tests/test_trained_gpu.py -m gpu -q -s > gpurun_out/final/trained_weights_parity.txt 2>&1; echo "trained rc=$?" >> gpurun_out/final/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final/summary.txt
bash scripts/final_profile.sh r05 > gpurun_out/final/final_profile.log 2>&1; echo "final_profile rc=$?" >> gpurun_out/final/summary.txt
PYTHONPATH=. python3 scripts/time_x3_wgrad.py > gpurun_out/final/x3_wgrad_table.txt 2>/dev/null
[ -f brats21_amd/libbrats_x3stamps.so ] && BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_x3stamps.so python3 scripts/probes/x3w_stamps.py > gpurun_out/final/x3_wgrad_stamps.txt 2>/dev/null
cat gpurun_out/final/summary.txt; tail -3 gpurun_out/final/pytest_gpu.log; tail -3 gpurun_out/final/smoke.log; tail -c 900 gpurun_out/final/bench.json
