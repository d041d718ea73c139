# usage: scripts/ab.sh "<shape>" ... ; compares brats21_amd/libbrats_hip_ab.so (A) with the current build (B) on one box
for sh in "$@"; do for rep in 1 2; do
  echo -n "A "; BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_hip_ab.so python scripts/time_conv.py $sh 1 20 | grep fwd
  echo -n "B "; python scripts/time_conv.py $sh 1 20 | grep fwd
done; done
