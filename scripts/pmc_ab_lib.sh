#!/bin/bash
# Same-box PMC passes over the dominant kernel for two builds of the CURRENT sources: bash scripts/pmc_ab_lib.sh <out> <libA> <libB>
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$1; mkdir -p $out; cd $root
for rep in 1 2; do
  for lib in $2 $3; do
    rm -rf gpurun_out/pmcab
    BRATS_HIP_LIB=$root/brats21_amd/$lib bash scripts/pmc.sh pmcab scripts/prof_conv.py dom > /dev/null 2>&1
    echo "== $lib rep $rep" >> $out/pmc_ab_lib.txt
    python3 scripts/pmc_report.py pmcab 2>&1 | grep -v "amdgpu.ids" | cut -c1-400 >> $out/pmc_ab_lib.txt
  done
done
cat $out/pmc_ab_lib.txt
