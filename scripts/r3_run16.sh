#!/bin/bash
# round 3, GPU run 16: EquiUnetASSPEvo: output head folded into the decoder1 block's backward -- tests, same-box A/B, kernel stats
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run16; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_assp_gpu.py -m gpu -x -q > $out/pytest_assp.log 2>&1; echo "pytest assp rc=$?" >> $out/summary.txt
timeout 1500 python -m pytest tests/test_headline_gpu.py tests/test_fp8_gpu.py -m gpu -x -q -k "assp" > $out/pytest_headline_assp.log 2>&1; echo "pytest headline assp rc=$?" >> $out/summary.txt
for rep in 1 2; do
  for fh in 0 1; do
    echo "== fold_head=$fh rep $rep" >> $out/ab.log
    BRATS_FOLD_HEAD=$fh timeout 600 python bench.py --model equiunet_assp_evo --graph --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg 2>>$out/ab.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['config']['loss'], r['roofline']['avg_ms'], r['roofline']['frac'])" >> $out/ab.log 2>&1
  done
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --model equiunet_assp_evo --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/assp_kernel_stats.csv; rm -rf $out/prof
cat $out/summary.txt; tail -4 $out/pytest_assp.log; tail -4 $out/pytest_headline_assp.log; cat $out/ab.log; grep -E "evonorm|head_bwd" $out/assp_kernel_stats.csv | cut -c1-70,200-400
