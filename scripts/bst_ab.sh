#!/bin/bash
# Backward statistics inside the input-gradient convolution (model.fold_bwd_stats) and the packed-pair epilogue statistics:
# tests, then same-box A/B through bench.py ("prev" = brats21_amd/libbrats_prev.so, the previous commit's library, if present).
#   usage: bash scripts/bst_ab.sh OUT [reps] [tests]
out=${1:-gpurun_out/bst}; reps=${2:-3}; mkdir -p $out
if [ "${3:-1}" != 0 ]; then
  timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_equiunet_gpu.py -q -x 2>&1 | tail -15 > $out/tests.log
  cat $out/tests.log
fi
cfgs=("hip BRATS_FOLD_BWD_STATS=0" "hip BRATS_FOLD_BWD_STATS=1")
[ -f brats21_amd/libbrats_prev.so ] && cfgs=("prev BRATS_FOLD_BWD_STATS=0" "${cfgs[@]}")
[ "$reps" -gt 0 ] && bash scripts/ab_bench.sh $out/ab $reps "${cfgs[@]}"
for fold in 0 1; do
  BRATS_FOLD_BWD_STATS=$fold timeout 600 python3 bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs --kernel-table 2>&1 | grep -i "conv_igemm\|gn_bwd\|tiles_finish\|chan_reduce" | cut -c1-200 > $out/kernels_fold$fold.txt
done
echo "== fold 0"; cat $out/kernels_fold0.txt; echo "== fold 1"; cat $out/kernels_fold1.txt
