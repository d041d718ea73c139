#!/bin/bash
# round 3, GPU run 1: loader-wave kernel correctness + same-process A/B, SLP on/off builds, configs[3] parity test
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run1; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_assp_gpu.py -m gpu -q -k "conv3d or se_gate or conv_evo_block or assp" > $out/pytest_conv.log 2>&1; echo "pytest conv rc=$?" >> $out/summary.txt
timeout 600 python scripts/time_ld.py 5 10 > $out/time_ld.log 2>&1
BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_hip_noslp.so timeout 600 python scripts/time_ld.py 5 10 > $out/time_ld_noslp.log 2>&1
for cfg in "ab 1" "hip 1" "hip_noslp 1" "hip 2" "hip_noslp 2" "ab 1" "hip 2"; do set -- $cfg
  f=$PWD/brats21_amd/libbrats_$1.so; [ "$1" = hip ] && f=$PWD/brats21_amd/libbrats_hip.so
  echo "== lib $1 mode $2" >> $out/bench_ab.log
  BRATS_HIP_LIB=$f BRATS_CONV_VS8=$2 timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['config']['loss'], r['roofline']['kernel'], r['roofline']['avg_ms'], r['roofline']['frac'], {k:v['ms_per_step'] for k,v in r['roofline']['families'].items()})" >> $out/bench_ab.log 2>&1
done
timeout 1500 python -m pytest tests/test_config3_gpu.py -m gpu -x -q -s > $out/pytest_config3.log 2>&1; echo "pytest config3 rc=$?" >> $out/summary.txt
tail -25 $out/pytest_conv.log; cat $out/time_ld.log; cat $out/time_ld_noslp.log; cat $out/bench_ab.log; tail -15 $out/pytest_config3.log
