#!/bin/bash
# rocprofv3 kernel statistics of the inference leg (configs[3]: 18 windows x 8 flips): bash scripts/prof_infer.sh <out-dir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_infer -- python3 scripts/prof_infer.py > $out/prof_infer.log 2>&1
cp $out/stats_infer/*/*kernel_stats.csv $out/infer_kernel_stats.csv
rm -rf $out/stats_infer
head -30 $out/infer_kernel_stats.csv | cut -c1-200; tail -2 $out/prof_infer.log | cut -c1-600
