#!/bin/bash
# Same-box A/B of library builds / switches through bench.py (replaces round 3's 26 one-off r3_run*.sh launchers: every one
# of them was this loop with other arguments).  Box-to-box spread of one binary is several per cent, so a change is only
# judged inside ONE gpurun call, alternating the configurations.
#   usage: scripts/ab_bench.sh OUT REPS "CFG" ["CFG" ...] [-- extra bench.py arguments]
#   CFG  = "label [ENV=VALUE ...]"; a label that names brats21_amd/libbrats_<label>.so selects that build (BRATS_HIP_LIB),
#          any other label runs the in-tree library; ENV=VALUE pairs are exported for that configuration only.
#   e.g.   scripts/ab_bench.sh gpurun_out/ab1 3 "hip" "hip_old" "hip BRATS_FOLD_POOL=0" -- --model equiunet_assp_evo
out=$1; reps=$2; shift 2
cfgs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done; [ "$1" = "--" ] && shift
extra=("$@")
mkdir -p $out
for rep in $(seq $reps); do for cfg in "${cfgs[@]}"; do
  set -- $cfg; label=$1; shift
  lib=$PWD/brats21_amd/libbrats_$label.so; envs=("$@"); [ -f "$lib" ] && [ "$label" != hip ] && envs+=("BRATS_HIP_LIB=$lib")
  echo -n "rep $rep | $cfg | " >> $out/ab.log
  env "${envs[@]}" timeout 900 python3 bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs "${extra[@]}" 2>>$out/ab.err | tail -1 | \
    python3 -c "import json,sys; r=json.loads(sys.stdin.read()); f=r['roofline'] or {}; print(r['ms_per_step'], 'ms  loss', r['config']['loss'], '| dominant', f.get('avg_ms'), 'ms frac', f.get('frac'), 'of box', f.get('frac_of_box'), '| box', (r.get('box') or {}).get('mfma_TFLOPs'), 'TF', (r.get('box') or {}).get('stream_TBps'), 'TB/s |', {k: v['ms_per_step'] for k, v in (f.get('families') or {}).items()})" >> $out/ab.log 2>&1
done; done
cat $out/ab.log
