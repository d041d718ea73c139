#!/bin/bash
# round 3, GPU run 4: epilogue diet (bias in accumulators, fma statistics, packed conversions) A/B against the previous build
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run4; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt
for rep in 1 2 3; do
  for lib in hip_noslp hip; do
    f=$PWD/brats21_amd/libbrats_$lib.so
    echo "== lib $lib rep $rep" >> $out/ab.log
    BRATS_HIP_LIB=$f timeout 300 python scripts/time_conv.py 48 48 128 1 20 2>>$out/ab.err | grep "fwd \|wgrad" >> $out/ab.log
    BRATS_HIP_LIB=$f timeout 300 python scripts/time_conv.py 48 96 128 1 20 2>>$out/ab.err | grep "fwd " >> $out/ab.log
    BRATS_HIP_LIB=$f timeout 300 python scripts/time_conv.py 96 96 64 1 20 2>>$out/ab.err | grep "fwd " >> $out/ab.log
    BRATS_HIP_LIB=$f timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'], {k:v['ms_per_step'] for k,v in r['roofline']['families'].items()})" >> $out/ab.log 2>&1
  done
done
BRATS_FORCE_DDP=1 timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg > $out/bench_forced_ddp.json 2>> $out/ab.err
cat $out/summary.txt; tail -4 $out/pytest_gpu.log; cat $out/ab.log; cut -c1-1200 $out/bench_forced_ddp.json; tail -5 $out/ab.err
