#!/bin/bash
# A/B builds of the library: one translation unit recompiled with extra -D flags, linked with the in-tree objects.
#   usage: scripts/build_variant.sh LABEL UNIT "FLAGS"      e.g.  scripts/build_variant.sh nw4 conv_wgrad "-DX3_NW=4"
# -> brats21_amd/libbrats_LABEL.so (git-ignored; selected with BRATS_HIP_LIB or scripts/ab_bench.sh "LABEL")
set -e
label=$1; unit=$2; flags=$3
cd "$(dirname "$0")/../brats21_amd/csrc"
make -j8 > /dev/null
CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -ffp-contract=off -fno-slp-vectorize"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc $CXXFLAGS $flags -c $unit.hip -o $tmp/$unit.o &
[ -f $unit.f16.o ] && /opt/rocm/bin/hipcc $CXXFLAGS $flags -DBRATS_FP16 -c $unit.hip -o $tmp/$unit.f16.o &
wait
objs=""
for o in *.o; do
  case $o in $unit.o|$unit.f16.o) objs="$objs $tmp/$o";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbrats_$label.so $objs
rm -rf $tmp; ls -la ../libbrats_$label.so
