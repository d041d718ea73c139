#!/bin/bash
# round 3, GPU run 18: output head forward on the raw convolution output (up1 never stored) -- tests, A/B, inference
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run18; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "groupnorm or head" > $out/pytest_ops.log 2>&1; echo "pytest ops rc=$?" >> $out/summary.txt
timeout 2400 python -m pytest tests/test_equiunet_gpu.py tests/test_headline_gpu.py tests/test_inference_gpu.py tests/test_fp8_gpu.py -m gpu -x -q > $out/pytest_net.log 2>&1; echo "pytest net rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for fh in 0 1; do
    echo "== fold_head_fwd=$fh rep $rep" >> $out/ab.log
    BRATS_FOLD_HEAD_FWD=$fh timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
  done
done
for fh in 0 1; do
  echo "== inference fold_head_fwd=$fh" >> $out/ab.log
  BRATS_FOLD_HEAD_FWD=$fh timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg --infer-headline-only 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print({k: v for k, v in r.get('inference', {}).items() if not isinstance(v, dict)})" >> $out/ab.log 2>&1
done
cat $out/summary.txt; tail -3 $out/pytest_ops.log; tail -3 $out/pytest_net.log; cat $out/ab.log; tail -3 $out/ab.err
