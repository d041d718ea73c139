#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2; do for lib in hip "$@"; do
  echo "=== $lib" >> $out/log.txt
  for cs in "48 128" "96 64" "24 128"; do
    BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_$lib.so python scripts/time_evo.py $cs 2>&1 | grep "TB/s" >> $out/log.txt
  done
done; done
cat $out/log.txt
