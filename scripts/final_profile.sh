#!/bin/bash
# Collects the round's committed evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command,
# per-layer conv table, and separate --pmc passes over the dominant conv shapes.  Outputs under gpurun_out/final/.
# usage: bash scripts/final_profile.sh [tag]     (tag = round prefix of the files copied to profiles/, e.g. r02)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
python3 bench.py --steps 20 --warmup 5 --kernel-table > $out/bench.json 2> $out/conv_table.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > $out/bench_profiled.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/bench_kernel_stats.csv
bash scripts/pmc.sh final/pmc scripts/prof_conv.py all > /dev/null 2>&1
python3 scripts/pmc_report.py final/pmc > $out/pmc_conv.txt
bash scripts/pmc.sh final/pmc_dom scripts/prof_conv.py dom > /dev/null 2>&1
python3 scripts/pmc_report.py final/pmc_dom > $out/pmc_dominant.txt
python3 scripts/pmc_report.py final/pmc_dom --json "conv_igemm cin=48 cout=48 k=3 dil=1 @2x128x128x128" "conv_igemm_vs8_kernel<24, 1, 3, false, false>" > $out/pmc_dominant.json
# round 4: the split-precision parity mode as the timed configuration (per-layer table), and the inference leg's kernel statistics
python3 bench.py --precision x3 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs --kernel-table > $out/bench_x3.json 2> $out/conv_table_x3.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x3 -- python3 bench.py --precision x3 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > /dev/null 2>&1
cp $out/stats_x3/*/*kernel_stats.csv $out/bench_x3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_infer -- python3 scripts/prof_infer.py > /dev/null 2>&1
cp $out/stats_infer/*/*kernel_stats.csv $out/infer_kernel_stats.csv
python3 bench.py --model equiunet_assp_evo --precision x3 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_assp_x3.json
python3 bench.py --steps 10 --warmup 3 --fp8 all --no-cpu-baseline 2> /dev/null | tail -1 > $out/bench_fp8.json
python3 bench.py --steps 10 --warmup 3 --fp8 all --graph --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_fp8_graph.json
python3 bench.py --model equiunet_assp_evo --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_assp.json
python3 bench.py --model equiunet_assp_evo --graph --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_assp_graph.json
python3 bench.py --model equiunet_assp_evo --width 64 --fp8 all --graph --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_assp64_fp8_graph.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_assp -- python3 bench.py --model equiunet_assp_evo --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > /dev/null 2>&1
cp $out/stats_assp/*/*kernel_stats.csv $out/bench_assp_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_assp64 -- python3 bench.py --model equiunet_assp_evo --width 64 --fp8 all --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > /dev/null 2>&1
cp $out/stats_assp64/*/*kernel_stats.csv $out/bench_assp64_fp8_kernel_stats.csv
PYTHONPATH=. python3 scripts/time_wgrad_f8.py > $out/wgrad_f8_table.txt 2>/dev/null
python3 bench.py --precision fp16 --steps 10 --warmup 3 --infer-headline-only --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_fp16.json
python3 bench.py --model equiunet_assp_evo --width 64 --precision fp16 --batch 4 --fp8 all --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_assp64_fp16_fp8_b4.json
BRATS_FORCE_DDP=1 python3 bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2> /dev/null | tail -1 > $out/bench_ddp1_forced.json
rm -rf $out/stats $out/stats_assp $out/stats_assp64 $out/stats_x3 $out/stats_infer
tail -c 600 $out/bench.json; echo; cat $out/pmc_conv.txt | cut -c1-250
