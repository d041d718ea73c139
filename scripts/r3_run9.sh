#!/bin/bash
# round 3, GPU run 9: consecutive tiles per workgroup for the Cout = 48 kernel (1 = default) -- correctness + A/B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run9; rm -rf $out; mkdir -p $out
for t in 2 3; do BRATS_CONV_VS8_TPW=$t timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3d" > $out/pytest_conv_tpw$t.log 2>&1; echo "pytest conv tpw=$t rc=$?" >> $out/summary.txt; done
for rep in 1 2 3; do
  for t in 1 2 4; do
    echo "== tpw=$t rep $rep" >> $out/ab.log
    BRATS_CONV_VS8_TPW=$t timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'], {k:v['ms_per_step'] for k,v in r['roofline']['families'].items()})" >> $out/ab.log 2>&1
  done
done
cat $out/summary.txt; cat $out/ab.log; tail -3 $out/ab.err
