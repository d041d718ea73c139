#!/bin/bash
# Weight packer through LDS (conv_host.hip pack_tile): layout tests, packing time of a step (one launch vs per layer) with the
# packer's kernel times from rocprofv3, then per-layer launches vs the one-launch plan through bench.py on the same box.
#   usage: bash scripts/pack_ab.sh OUT [reps]
out=${1:-gpurun_out/pack}; reps=${2:-3}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "pack_weights or conv3d_fwd_dgrad_wgrad or channel_slice" 2>&1 | tail -3 > $out/tests.log
timeout 900 python3 -m pytest tests/test_equiunet_gpu.py tests/test_x3_gpu.py tests/test_assp_gpu.py -q -x 2>&1 | tail -3 >> $out/tests.log
cat $out/tests.log
timeout 300 python3 scripts/time_pack.py 2>&1 | grep -v Warn | tee $out/time_pack.txt
timeout 300 python3 scripts/time_pack.py equiunet_assp_evo 2>&1 | grep -v Warn | tee -a $out/time_pack.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 scripts/time_pack.py > /dev/null 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -i "pack_weights\|Name" "$f" | tee -a $out/time_pack.txt
[ "$reps" -gt 0 ] && bash scripts/ab_bench.sh $out/ab $reps "hip BRATS_PACK_PLAN=0" "hip BRATS_PACK_PLAN=1"
