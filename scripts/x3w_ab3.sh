#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2; do for lib in hip "$@"; do
  echo "=== $lib" >> $out/log.txt
  BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_$lib.so timeout 600 python scripts/time_x3_wgrad.py 2>&1 | grep "128^3\|64^3: " >> $out/log.txt
done; done
cat $out/log.txt
