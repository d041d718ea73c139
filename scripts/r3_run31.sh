#!/bin/bash
# round 3, GPU run 31: vectorised input layout pass -- full GPU suite, step, inference, kernel stats
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run31; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt
for rep in 1 2; do
  timeout 600 python bench.py --steps 30 --warmup 10 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('equiunet step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
done
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-leg --infer-headline-only 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print({k: v for k, v in r.get('inference', {}).items() if not isinstance(v, (dict, str))})" >> $out/ab.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv; rm -rf $out/prof
cat $out/summary.txt; grep -n "passed\|failed" $out/pytest_gpu.log | tail -2; cat $out/ab.log | cut -c1-200; grep -E "ncdhw" $out/kernel_stats.csv | cut -c1-200
