#!/bin/bash
# round 6, fourth GPU pass: what changed since the third (first-layer wgrad, PRE + 8-wave, sweep), then the trained tests
mkdir -p gpurun_out
python scripts/time_first_wgrad.py > gpurun_out/r06_first_wgrad_ab.txt 2>&1
BRATS_WGRAD_ALLTAPS=1 python scripts/time_first_wgrad.py >> gpurun_out/r06_first_wgrad_ab.txt 2>&1
cat gpurun_out/r06_first_wgrad_ab.txt
python -m pytest tests/test_ops_gpu.py tests/test_equiunet_gpu.py tests/test_bench_gpu.py tests/test_headline_gpu.py -m gpu -q -k "first_layer or normalise_on_load or k_parity or bench or headline" > gpurun_out/r06_check4_a.txt 2>&1; tail -8 gpurun_out/r06_check4_a.txt
python -m pytest tests/test_trained_gpu.py -m gpu -q -s > gpurun_out/r06_check4_trained.txt 2>&1; tail -5 gpurun_out/r06_check4_trained.txt
grep -h "weight sets x\|volumes  \|not counted" gpurun_out/r06_check4_trained.txt gpurun_out/r06_trained_sweep_*.txt | sort -u
