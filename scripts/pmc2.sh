#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/$tag/a -- python "$@" > gpurun_out/$tag/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/$tag/b -- python "$@" > gpurun_out/$tag/b.log 2>&1
python - <<PY
import csv,glob,collections
for part in "ab":
    for f in glob.glob("gpurun_out/$tag/%s/*/*counter_collection.csv"%part):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "wgrad_kernel" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,d in agg.items():
            print(k, {c: "%.3g"%(sum(v)/len(v)) for c,v in d.items()})
PY
