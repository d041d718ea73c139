#!/bin/bash
# rocprofv3 kernel statistics of the inference leg with and without normalise-on-load: bash scripts/prof_infer_ab.sh <out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
export BRATS_NORM_ON_LOAD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s0 -- python3 scripts/prof_infer.py > $out/prof_infer_off.log 2>&1
cp $out/s0/*/*kernel_stats.csv $out/infer_kernel_stats_two_pass.csv; rm -rf $out/s0
export BRATS_NORM_ON_LOAD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s1 -- python3 scripts/prof_infer.py > $out/prof_infer_on.log 2>&1
cp $out/s1/*/*kernel_stats.csv $out/infer_kernel_stats_norm_on_load.csv; rm -rf $out/s1
for f in two_pass norm_on_load; do echo "== $f"; head -12 $out/infer_kernel_stats_$f.csv | cut -c1-150; done
