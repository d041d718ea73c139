#!/bin/bash
# round 3, GPU run 19: deep-supervision heads on a second stream (opt-in) -- tests with the switch on, same-box A/B, graph
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run19; rm -rf $out; mkdir -p $out
BRATS_HEADS_STREAM=1 timeout 2400 python -m pytest tests/test_equiunet_gpu.py tests/test_headline_gpu.py -m gpu -x -q -k "not assp" > $out/pytest_net.log 2>&1; echo "pytest net (heads stream) rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for hs in 0 1; do
    echo "== heads_stream=$hs rep $rep" >> $out/ab.log
    BRATS_HEADS_STREAM=$hs timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
  done
done
for hs in 0 1; do
  echo "== graph heads_stream=$hs" >> $out/ab.log
  BRATS_HEADS_STREAM=$hs timeout 600 python bench.py --graph --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'])" >> $out/ab.log 2>&1
done
cat $out/summary.txt; tail -3 $out/pytest_net.log; cat $out/ab.log; tail -3 $out/ab.err
