#!/bin/bash
# round 3, GPU run 8: ablation builds of the dominant kernel and of the loader-wave kernel (diagnostic libraries, wrong results
# by construction): which memory stream -- weight fragments inside the MMA loop, halo loads, epilogue stores -- costs what
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run8; rm -rf $out; mkdir -p $out
for rep in 1 2 3; do
  for lib in hip hip_abl_NOWLOAD hip_abl_NOHALO hip_abl_NOSTORE hip_abl_NOLOADS hip_abl_ALL; do
    f=$PWD/brats21_amd/libbrats_$lib.so
    for mode in "1 0" "2 1" "2 0"; do set -- $mode
      echo "== $lib mode$1 v$2 rep $rep" >> $out/abl.log
      BRATS_CONV_VS8=$1 BRATS_CONV_LD_VARIANT=$2 BRATS_HIP_LIB=$f timeout 300 python scripts/time_conv.py 48 48 128 1 20 2>>$out/abl.err | grep "fwd " >> $out/abl.log
    done
  done
done
cat $out/abl.log | tail -120; tail -3 $out/abl.err
