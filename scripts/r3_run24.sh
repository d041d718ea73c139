#!/bin/bash
# round 3, GPU run 24: pooling backward + skip add inside the GroupNorm backward of the level's last layer -- tests, A/B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run24; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "pool or groupnorm or head" > $out/pytest_ops.log 2>&1; echo "pytest ops rc=$?" >> $out/summary.txt
timeout 2400 python -m pytest tests/test_equiunet_gpu.py tests/test_headline_gpu.py tests/test_fp8_gpu.py tests/test_ddp_gpu.py -m gpu -x -q > $out/pytest_net.log 2>&1; echo "pytest net rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for fp in 0 1; do
    echo "== fold_pool=$fp rep $rep" >> $out/ab.log
    BRATS_FOLD_POOL=$fp timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
  done
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv; rm -rf $out/prof
cat $out/summary.txt; tail -3 $out/pytest_ops.log; tail -3 $out/pytest_net.log; cat $out/ab.log; grep -E "gn_bwd|maxpool" $out/kernel_stats.csv | cut -c1-75,180-330
