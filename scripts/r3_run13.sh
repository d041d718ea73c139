#!/bin/bash
# round 3, GPU run 13: EvoNorm + SE forward without storing z (statistics pass, gate, scaled EvoNorm pass) on top of run 12
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run13; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_assp_gpu.py -m gpu -x -q > $out/pytest_assp.log 2>&1; echo "pytest assp rc=$?" >> $out/summary.txt
timeout 1500 python -m pytest tests/test_headline_gpu.py tests/test_infer_gpu.py -m gpu -x -q -k "assp" > $out/pytest_headline_assp.log 2>&1; echo "pytest headline assp rc=$?" >> $out/summary.txt
for rep in 1 2; do
  timeout 600 python bench.py --model equiunet_assp_evo --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/bench.err | tail -1 | cut -c1-260 >> $out/bench.log
  timeout 600 python bench.py --model equiunet_assp_evo --graph --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/bench.err | tail -1 | cut -c1-260 >> $out/bench.log
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --model equiunet_assp_evo --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/assp_kernel_stats.csv; rm -rf $out/prof
cat $out/summary.txt; tail -5 $out/pytest_assp.log; tail -5 $out/pytest_headline_assp.log; cat $out/bench.log; head -40 $out/assp_kernel_stats.csv | cut -c1-150
