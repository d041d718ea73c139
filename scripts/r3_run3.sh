#!/bin/bash
# round 3, GPU run 3: fp16 twin build -- op tests in three dtypes, headline fp16 tests, full suite, bench --precision fp16
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run3; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > $out/pytest_ops.log 2>&1; echo "pytest ops rc=$?" >> $out/summary.txt
timeout 1500 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -s -k "fp16" > $out/pytest_fp16.log 2>&1; echo "pytest fp16 rc=$?" >> $out/summary.txt
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_ops_gpu.py > $out/pytest_rest.log 2>&1; echo "pytest rest rc=$?" >> $out/summary.txt
timeout 600 python bench.py --precision fp16 --steps 20 --warmup 5 --infer-headline-only --no-cpu-baseline --no-parity-leg > $out/bench_fp16.json 2> $out/bench_fp16.err; echo "bench fp16 rc=$?" >> $out/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg > $out/bench_bf16.json 2> $out/bench_bf16.err
cat $out/summary.txt; tail -15 $out/pytest_ops.log; tail -30 $out/pytest_fp16.log; tail -15 $out/pytest_rest.log; cat $out/bench_fp16.json | cut -c1-600; tail -5 $out/bench_fp16.err; cat $out/bench_bf16.json | cut -c1-300
