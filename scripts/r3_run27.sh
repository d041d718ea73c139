#!/bin/bash
# round 3, GPU run 27: split-K slab reduction of the weight gradient with eight slabs in flight per thread -- tests, step, stats
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run27; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -m gpu -x -q -k "wgrad or conv" > $out/pytest_ops.log 2>&1; echo "pytest ops rc=$?" >> $out/summary.txt
timeout 2400 python -m pytest tests/test_equiunet_gpu.py tests/test_assp_gpu.py tests/test_headline_gpu.py -m gpu -x -q > $out/pytest_net.log 2>&1; echo "pytest net rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('equiunet step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
done
BRATS_HIP_LIB=$GRAFT_REPO_ROOT/brats21_amd/libbrats_hip_ab.so true
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv; rm -rf $out/prof
cat $out/summary.txt; tail -3 $out/pytest_ops.log; tail -3 $out/pytest_net.log; cat $out/ab.log | cut -c1-200; grep -E "wgrad_reduce|ordered_sum|gn_bwd_finish|gn_chan" $out/kernel_stats.csv | cut -c1-200
