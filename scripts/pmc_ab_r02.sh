#!/bin/bash
# Same-box PMC passes over the dominant kernel (48 -> 48 @2x128^3 forward): the round-2 final tree against the current tree.
# Prepare the old tree first (it is not kept in the repository; scratch/ is git-ignored but travels with gpurun):
#   git worktree add scratch/r02tree f8e7ebe && make -C scratch/r02tree/brats21_amd/csrc -j8
# (result of the round-4 run: profiles/r04_fetch_r02_tree_vs_current_same_box.txt) -- why did FETCH_SIZE read 584 MiB in r02 and 682 MiB in r03?  usage: bash scripts/pmc_ab_r02.sh <out>
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$1; mkdir -p $out
for rep in 1 2; do
  for tree in r02 cur; do
    if [ $tree = r02 ]; then export GRAFT_REPO_ROOT=$root/scratch/r02tree; else export GRAFT_REPO_ROOT=$root; fi
    cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmcab
    bash scripts/pmc.sh pmcab scripts/prof_conv.py dom > /dev/null 2>&1
    echo "== $tree rep $rep" >> $out/pmc_ab.txt
    python3 scripts/pmc_report.py pmcab 2>&1 | grep -v "amdgpu.ids" | cut -c1-400 >> $out/pmc_ab.txt
  done
done
export GRAFT_REPO_ROOT=$root; cd $root
cat $out/pmc_ab.txt
