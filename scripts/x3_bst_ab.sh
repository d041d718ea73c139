#!/bin/bash
# parity mode (x3): the backward-statistics fold of the split-precision input gradient -- tests, then bench.py --precision x3 for
# both networks with BRATS_FOLD_BWD_STATS = 0 / 1 alternating on the same box;  bash scripts/x3_bst_ab.sh LABEL
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -k "backward_statistics or backward_arithmetic or gradients_vs_f64" 2>&1 | grep -v "^$" | tail -25 >> $out/log.txt
timeout 900 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -s -k "vs_oracle_f32_and_bf16" 2>&1 | grep "bf16:\|passed\|failed" >> $out/log.txt
for model in equiunet equiunet_assp_evo; do
for f in 0 1 0 1 0 1; do
  echo "=== bench x3 $model, BRATS_FOLD_BWD_STATS=$f" >> $out/log.txt
  BRATS_FOLD_BWD_STATS=$f python3 bench.py --model $model --precision x3 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['loss'], r['roofline']['families'])" >> $out/log.txt 2>&1
done
done
cat $out/log.txt
