#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
timeout 900 python -m pytest tests/test_x3_gpu.py -m gpu -x -q -s -k "fused" 2>&1 | grep -E "passed|failed|Error|error" | tail -4 >> $out/log.txt
for rep in 1 2; do for lib in hip "$@"; do
  echo "=== $lib" >> $out/log.txt
  BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_$lib.so timeout 600 python scripts/time_x3_wgrad.py 2>&1 | grep "\^3: " >> $out/log.txt
done; done
cat $out/log.txt
