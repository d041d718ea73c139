#!/bin/bash
# round 3, GPU run 17: EquiUnetASSPEvo same-box A/B of the folded output head (graph replay)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run17; rm -rf $out; mkdir -p $out
for rep in 1 2 3; do
  for fh in 0 1; do
    echo "== fold_head=$fh rep $rep" >> $out/ab.log
    BRATS_FOLD_HEAD=$fh timeout 600 python bench.py --model equiunet_assp_evo --graph --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'])" >> $out/ab.log 2>&1
  done
done
BRATS_FOLD_HEAD=1 timeout 600 python bench.py --model equiunet_assp_evo --width 64 --fp8 all --graph --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | cut -c1-200 >> $out/ab.log
cat $out/ab.log
