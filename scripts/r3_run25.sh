#!/bin/bash
# round 3, GPU run 25: EquiUnetASSPEvo pooling backward + bridge-gradient add inside the block's EvoNorm / SE backward -- tests, A/B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run25; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_assp_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "pool or assp or evonorm or se_ or groupnorm" > $out/pytest_assp.log 2>&1; echo "pytest assp rc=$?" >> $out/summary.txt
timeout 2400 python -m pytest tests/test_headline_gpu.py tests/test_fp8_gpu.py tests/test_equiunet_gpu.py -m gpu -x -q > $out/pytest_net.log 2>&1; echo "pytest net rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for fp in 0 1; do
    echo "== fold_pool=$fp rep $rep" >> $out/ab.log
    BRATS_FOLD_POOL=$fp timeout 600 python bench.py --model equiunet_assp_evo --graph --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('assp graph step', r['ms_per_step'], r['config']['loss'])" >> $out/ab.log 2>&1
  done
done
cat $out/summary.txt; tail -3 $out/pytest_assp.log; tail -3 $out/pytest_net.log; cat $out/ab.log
