#!/bin/bash
# Same-box A/B of two builds of libbrats_hip.so (box-to-box spread of one binary is +-3 %, so small changes are judged
# here).  Copy the baseline build to brats21_amd/libbrats_hip_ab.so (= A), rebuild (= B), then on the GPU box:
#   scripts/ab.sh "fwd " "48 48 128" "96 48 128"    # forward igemm of cin cout size [dil]
#   scripts/ab.sh wgrad "48 48 128 1"               # weight gradient
kind=$1; shift
for sh in "$@"; do for rep in 1 2 3; do
  a=$(BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_hip_ab.so python scripts/time_conv.py $sh 1 20 2>/dev/null | grep "$kind")
  b=$(python scripts/time_conv.py $sh 1 20 2>/dev/null | grep "$kind")
  echo "A: $a"; echo "B: $b"
done; done
