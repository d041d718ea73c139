#!/bin/bash
# x3 parity-mode step: tests + bench.py --precision x3 with the fused weight gradient on / off (same box)
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests/test_x3_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $out/log.txt
for f in 0 1 0 1; do
  echo "=== bench x3, BRATS_X3_WGRAD_FUSED=$f" >> $out/log.txt
  BRATS_X3_WGRAD_FUSED=$f python3 bench.py --precision x3 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['loss'], r['roofline']['families'])" >> $out/log.txt 2>&1
done
BRATS_X3_WGRAD_FUSED=1 python3 bench.py --precision x3 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs --kernel-table 2>&1 >/dev/null | grep wgrad >> $out/log.txt
cat $out/log.txt
