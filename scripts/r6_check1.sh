#!/bin/bash
# round 6, first GPU pass: the changed host paths (DDP over real RCCL at world 1, bench rehearsal line, optimizer / bcn / dropout fixes)
mkdir -p gpurun_out
python -m pytest tests/test_ddp_gpu.py tests/test_bench_gpu.py tests/test_optim_gpu.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r06_check1_pytest.txt
python -m pytest tests/test_assp_gpu.py tests/test_equiunet_gpu.py -m gpu -x -q -s -k "large_beta or bcn or dropout" 2>&1 | tail -30 >> gpurun_out/r06_check1_pytest.txt
python scripts/host_time.py > gpurun_out/r06_host_time.txt 2>&1
BRATS_FORCE_DDP=rccl python bench.py --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > gpurun_out/r06_rehearsal_line.json 2> gpurun_out/r06_rehearsal_err.txt
BRATS_FORCE_DDP=rccl python bench.py --model equiunet_assp_evo --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > gpurun_out/r06_rehearsal_assp_line.json 2>> gpurun_out/r06_rehearsal_err.txt
cat gpurun_out/r06_check1_pytest.txt gpurun_out/r06_host_time.txt; tail -5 gpurun_out/r06_rehearsal_err.txt
