#!/bin/bash
# fp16 loop (the reference's autocast + GradScaler arithmetic): Ranger2020 under the GradScaler protocol (device-side skip, no unscale
# pass) against GradScaler's own host check + unscale pass, same box;  bash scripts/fp16_amp_ab.sh LABEL
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_headline_gpu.py -m gpu -x -q -k "gradscaler or ranger" 2>&1 | tail -3 >> $out/log.txt
for rep in 1 2 3; do for v in 0 1; do
  echo -n "rep $rep BRATS_RANGER_AMP=$v fp16: " >> $out/log.txt
  BRATS_RANGER_AMP=$v python3 bench.py --precision fp16 --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['loss'])" >> $out/log.txt 2>&1
done; echo -n "rep $rep bf16: " >> $out/log.txt
  python3 bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['loss'])" >> $out/log.txt 2>&1
done
cat $out/log.txt
