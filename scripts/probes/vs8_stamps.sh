#!/bin/bash
# Diagnostic build of the 4x8x16-tile implicit-GEMM kernel with s_memtime stamps (never part of libbrats_hip.so):
#   bash scripts/probes/vs8_stamps.sh ; then on the GPU box: python scripts/probes/vs8_stamps.py
set -e
cd "$(dirname "$0")/../../brats21_amd/csrc"
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -ffp-contract=off -DBRATS_VS8_STAMPS $EXTRA"   # EXTRA=-DBRATS_VS8_FAKEW: L1-resident weights ablation
/opt/rocm/bin/hipcc $F -c conv_bf16_k3_d1.hip -o /tmp/vs8_a.o &
/opt/rocm/bin/hipcc $F -c conv_host.hip -o /tmp/vs8_b.o &
wait
objs=$(ls *.o | grep -v "conv_bf16_k3_d1.o\|conv_host.o\|conv_bf16_k3_d2.o\|conv_bf16_k1_d1.o\|conv_f32")
# (the other conv_* units share ConvParams: rebuild them with the same define so that the struct layouts agree)
for u in conv_bf16_k3_d2 conv_bf16_k1_d1 conv_f32_k1_d1 conv_f32_k3_d1 conv_f32_k3_d2 conv_f8_host conv_f8_k3_d1 conv_f8_k3_d2; do
  /opt/rocm/bin/hipcc $F -c $u.hip -o /tmp/vs8_$u.o &
done
wait
objs=$(ls *.o | grep -v "^conv_bf16\|^conv_f32\|^conv_f8\|^conv_host")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbrats_hip_stamps8${SUFFIX}.so $objs /tmp/vs8_*.o
ls -la ../libbrats_hip_stamps8${SUFFIX}.so
