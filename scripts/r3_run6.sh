#!/bin/bash
# round 3, GPU run 6: LDS-staged weight packing (per-layer and multi-tensor) -- layout oracle test, suite, A/B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run6; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "pack_weights or conv3d" > $out/pytest_pack.log 2>&1; echo "pytest pack rc=$?" >> $out/summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 1" "0 1"; do set -- $cfg
    echo "== PACK_LDS=$1 PACK_PLAN=$2 rep $rep" >> $out/ab.log
    BRATS_PACK_LDS=$1 BRATS_PACK_PLAN=$2 timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BRATS_BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg > $out/prof.log 2>&1
f=$(ls -S $out/prof/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv
cat $out/summary.txt; tail -3 $out/pytest_pack.log; tail -3 $out/pytest_gpu.log; cat $out/ab.log; head -30 $out/kernel_stats.csv | cut -c1-200
