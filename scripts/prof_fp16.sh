cd $GRAFT_REPO_ROOT; out=gpurun_out/fp16prof; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --precision fp16 --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs > $out/bench.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/fp16_kernel_stats.csv
for p in fp16 bf16 fp16 bf16; do python3 bench.py --precision $p --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$p', r['ms_per_step'])"; done > $out/times.txt
cat $out/times.txt
