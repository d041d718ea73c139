for sh in "$@"; do for rep in 1 2; do
  echo -n "A "; BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_hip_ab.so python scripts/time_conv.py $sh 20 | grep wgrad
  echo -n "B "; python scripts/time_conv.py $sh 20 | grep wgrad
done; done
