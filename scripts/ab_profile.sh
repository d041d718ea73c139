#!/bin/bash
# Same-box per-kernel comparison of two builds: brats21_amd/libbrats_hip_ab.so (A) against the in-tree library (B).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/abprof; rm -rf $out; mkdir -p $out
BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_hip_ab.so BRATS_BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/A -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > $out/A.log 2>&1
BRATS_BENCH_NO_TIMER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/B -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > $out/B.log 2>&1
python3 - <<'PY'
import csv, glob, os
def load(d):
    f = max(glob.glob(f"gpurun_out/abprof/{d}/*/*kernel_stats.csv"), key=os.path.getsize)
    return {r["Name"][:70]: (float(r["TotalDurationNs"]) / 13e6, int(r["Calls"]) / 13) for r in csv.DictReader(open(f))}
A, B = load("A"), load("B")
names = sorted(set(A) | set(B), key=lambda n: -max(A.get(n, (0, 0))[0], B.get(n, (0, 0))[0]))
print(f"{'kernel':70s} {'A ms':>8s} {'B ms':>8s} {'B-A':>8s}")
for n in names[:45]:
    a, b = A.get(n, (0, 0))[0], B.get(n, (0, 0))[0]
    print(f"{n:70s} {a:8.3f} {b:8.3f} {b - a:+8.3f}")
print("total", sum(v[0] for v in A.values()), sum(v[0] for v in B.values()))
PY
