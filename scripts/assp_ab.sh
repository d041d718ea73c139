#!/bin/bash
# EquiUnetASSPEvo-48 (BASELINE.json configs[2], one rank's share): the EvoNorm backward-statistics fold on / off, same box
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 900 python -m pytest tests/test_assp_gpu.py -m gpu -x -q -k "fold" 2>&1 | tail -2 >> $out/log.txt
for rep in 1 2 3; do for f in 0 1; do
  echo -n "rep $rep BRATS_FOLD_BWD_STATS=$f: " >> $out/log.txt
  BRATS_FOLD_BWD_STATS=$f python3 bench.py --model equiunet_assp_evo --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['loss'], r['roofline']['families'])" >> $out/log.txt 2>&1
done; done
cat $out/log.txt
