#!/bin/bash
# round 3, GPU run 2: full GPU suite on the no-SLP build + new bench legs + ASSP eager / graph with the native SE gate
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run2; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
timeout 600 python bench.py --model equiunet_assp_evo --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg > $out/bench_assp.json 2> $out/bench_assp.err
timeout 600 python bench.py --model equiunet_assp_evo --graph --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg > $out/bench_assp_graph.json 2> $out/bench_assp_graph.err
BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_hip_ab.so timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg > $out/bench_r2lib.json 2> $out/bench_r2lib.err
tail -5 $out/pytest_gpu.log; cat $out/summary.txt; cat $out/bench.json; tail -3 $out/bench.err; cat $out/bench_assp.json $out/bench_assp_graph.json; cat $out/bench_r2lib.json; tail -5 $out/bench_r2lib.err
