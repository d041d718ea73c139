#!/bin/bash
# fused x3 weight gradient: correctness (pytest) + kernel times for the in-tree build and A/B builds
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
for lib in hip "$@"; do
  export BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_$lib.so
  echo "=== $lib" >> $out/log.txt
  timeout 900 python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -k "fused" 2>&1 | grep -E "wgrad|passed|failed|Error|error" | tail -14 >> $out/log.txt
  timeout 600 python scripts/time_x3_wgrad.py >> $out/log.txt 2>&1
done
cat $out/log.txt
