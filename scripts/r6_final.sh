#!/bin/bash
# round 6, final GPU run on the final tree: whole GPU suite (+ the trained-weights tests with their printed numbers), smoke, the bench
# line (live PMC traffic inside), rocprofv3 kernel statistics of the same command, the separate --pmc passes over the dominant
# kernel, this round's same-box A/B tables, the N > 1 rehearsal lines.  Outputs under gpurun_out/final/, harvested by
# scripts/harvest_final.sh r06 into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out; rm -f $out/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_trained_gpu.py > $out/pytest_gpu.log 2>&1; echo "pytest gpu (all but the trained-weights file) rc=$?" >> $out/summary.txt
BRATS_SWEEP_FULL=1 timeout 2400 python -m pytest tests/test_trained_gpu.py -m gpu -q -s > $out/trained_weights_parity.txt 2>&1; echo "trained rc=$?" >> $out/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/summary.txt
python3 bench.py --kernel-table > $out/bench.json 2> $out/conv_table.txt; echo "bench rc=$?" >> $out/summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs --no-pmc-traffic > $out/bench_profiled.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/bench_kernel_stats.csv; rm -rf $out/stats
bash scripts/pmc.sh final/pmc_dom scripts/prof_conv.py dom > /dev/null 2>&1
python3 scripts/pmc_report.py final/pmc_dom > $out/pmc_dominant.txt
python3 scripts/pmc_report.py final/pmc_dom --json "conv_igemm cin=48 cout=48 k=3 dil=1 @2x128x128x128" "conv_igemm_vs8_kernel<24, 1, 3, false, false>" > $out/pmc_dominant.json
python3 scripts/time_kp.py > $out/kp_ab.txt 2>/dev/null
python3 scripts/time_first.py > $out/first_layer_ab.txt 2>/dev/null
BRATS_CONV_FIRST=0 python3 scripts/time_first.py 2>/dev/null | sed 's/persistent (dense out) /tile kernel, dense out/' | grep dense >> $out/first_layer_ab.txt
python3 scripts/time_first_wgrad.py > $out/first_layer_wgrad_ab.txt 2>/dev/null
BRATS_WGRAD_ALLTAPS=1 python3 scripts/time_first_wgrad.py 2>/dev/null | sed 's/first-layer wgrad/first-layer wgrad, register-staging form (BRATS_WGRAD_ALLTAPS=1)/' >> $out/first_layer_wgrad_ab.txt
python3 scripts/host_time.py > $out/host_time.txt 2>/dev/null
BRATS_FORCE_DDP=rccl python3 bench.py --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs --no-pmc-traffic 2> /dev/null | tail -1 > $out/rehearsal_equiunet.json
BRATS_FORCE_DDP=rccl python3 bench.py --model equiunet_assp_evo --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/rehearsal_assp.json
python3 bench.py --model equiunet_assp_evo --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs --kernel-table 2> $out/conv_table_assp.txt | tail -1 > $out/bench_assp.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_infer -- python3 scripts/prof_infer.py > /dev/null 2>&1
cp $out/stats_infer/*/*kernel_stats.csv $out/infer_kernel_stats.csv; rm -rf $out/stats_infer
cat $out/summary.txt; tail -3 $out/pytest_gpu.log; tail -3 $out/smoke.log; tail -c 1500 $out/bench.json
