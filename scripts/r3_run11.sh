#!/bin/bash
# round 3, GPU run 11: deep heads folded into the fused Dice passes -- equivalence test, full suite, same-box A/B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run11; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_equiunet_gpu.py -m gpu -x -q -k "lazy" > $out/pytest_lazy.log 2>&1; echo "pytest lazy rc=$?" >> $out/summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for lz in 0 1; do
    echo "== lazy=$lz rep $rep" >> $out/ab.log
    BRATS_LAZY_HEADS=$lz timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
  done
done
BRATS_LAZY_HEADS=1 timeout 600 python bench.py --model equiunet_assp_evo --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | cut -c1-200 >> $out/ab.log
BRATS_LAZY_HEADS=0 timeout 600 python bench.py --model equiunet_assp_evo --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | cut -c1-200 >> $out/ab.log
cat $out/summary.txt; tail -4 $out/pytest_lazy.log; tail -4 $out/pytest_gpu.log; cat $out/ab.log; tail -3 $out/ab.err
