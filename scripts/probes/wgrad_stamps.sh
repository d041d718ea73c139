#!/bin/bash
# Diagnostic build of the all-taps weight-gradient kernel with s_memtime stamps (never part of libbrats_hip.so):
#   bash scripts/probes/wgrad_stamps.sh        (here: builds brats21_amd/libbrats_hip_stamps.so)
#   python scripts/probes/wgrad_stamps.py      (GPU box: per-phase cycle shares of a tile)
set -e
cd "$(dirname "$0")/../../brats21_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -ffp-contract=off -DBRATS_WGRAD_STAMPS -c conv_wgrad.hip -o /tmp/conv_wgrad_stamps.o
objs=$(ls *.o | grep -v conv_wgrad.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbrats_hip_stamps.so $objs /tmp/conv_wgrad_stamps.o
ls -la ../libbrats_hip_stamps.so
