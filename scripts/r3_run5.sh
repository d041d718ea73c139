#!/bin/bash
# round 3, GPU run 5: three workgroups per CU for the Cout = 48 kernel (168-register build) against the default
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run5; mkdir -p $out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3d" > $out/pytest_conv.log 2>&1; echo "pytest conv rc=$?" >> $out/summary.txt
BRATS_CONV_VS8_W3=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv3d" > $out/pytest_conv_w3.log 2>&1; echo "pytest conv w3 rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for w3 in 0 1; do
    echo "== w3=$w3 rep $rep" >> $out/ab.log
    BRATS_CONV_VS8_W3=$w3 timeout 300 python scripts/time_conv.py 48 48 128 1 20 2>>$out/ab.err | grep "fwd " >> $out/ab.log
    BRATS_CONV_VS8_W3=$w3 timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'], {k:v['ms_per_step'] for k,v in r['roofline']['families'].items()})" >> $out/ab.log 2>&1
  done
done
cat $out/summary.txt; tail -3 $out/pytest_conv_w3.log; cat $out/ab.log; tail -3 $out/ab.err
