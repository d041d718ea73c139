#!/bin/bash
# Diagnostic build of the WHOLE library with -DBRATS_VS8_STAMPS (ConvParams grows a member, so every unit and both 16-bit twins are
# rebuilt; never part of libbrats_hip.so): bash scripts/probes/stamps_build.sh -> brats21_amd/libbrats_diag.so
# then on the GPU box: python scripts/probes/igemm_stamps.py 384 384 16 1 1   (conv_igemm_kernel)   or   python scripts/probes/vs8_stamps.py   (the 4x8x16-tile kernel)
# (delete the library afterwards: it is 13 MB that would travel with every gpurun call)
set -e
cd "$(dirname "$0")/../../brats21_amd/csrc"
make -j8 twin_dispatch.o > /dev/null
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -ffp-contract=off -fno-slp-vectorize -DBRATS_VS8_STAMPS"
tmp=$(mktemp -d)
TWIN="conv_host conv_f8_host conv_wgrad dconv dropout head layout norm pool_up conv_bf16_k1_d1 conv_bf16_k3_d1 conv_bf16_k3_d2 conv_f8_k3_d1 conv_f8_k3_d2 conv_x3_k3_d1 conv_x3_k3_d2 conv_bf16_k3_pre conv_bf16_k3_bst conv_x3_k3_bst"
n=0
for f in *.hip; do
  u=${f%.hip}
  /opt/rocm/bin/hipcc $F -c $f -o $tmp/$u.o &
  n=$((n+1)); [ $((n % 8)) -eq 0 ] && wait
done
wait
for u in $TWIN; do
  /opt/rocm/bin/hipcc $F -DBRATS_FP16 -c $u.hip -o $tmp/$u.f16.o &
  n=$((n+1)); [ $((n % 8)) -eq 0 ] && wait
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbrats_diag.so $tmp/*.o twin_dispatch.o
rm -rf $tmp; ls -la ../libbrats_diag.so
