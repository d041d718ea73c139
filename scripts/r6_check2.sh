#!/bin/bash
# round 6, second GPU pass: whole GPU suite (incl. the trained-weights sweep), K-parity timing, a default bench line
mkdir -p gpurun_out
python scripts/time_kp.py > gpurun_out/r06_kp_ab.txt 2>&1
python scripts/time_first.py >> gpurun_out/r06_kp_ab.txt 2>&1
python -m pytest tests -m gpu -x -q -s 2>&1 > gpurun_out/r06_check2_pytest_full.txt; echo "pytest rc=$?" >> gpurun_out/r06_check2_pytest_full.txt
tail -25 gpurun_out/r06_check2_pytest_full.txt
python bench.py --kernel-table > gpurun_out/r06_check2_bench.json 2> gpurun_out/r06_check2_bench_err.txt
cat gpurun_out/r06_kp_ab.txt; grep "^#" gpurun_out/r06_check2_bench_err.txt | head -60
python -c "
import json;r=json.loads(open('gpurun_out/r06_check2_bench.json').read().strip().splitlines()[-1]);print(r['value'],r['ms_per_step'],r['host_enqueue_ms'],r['roofline']['frac'],r['roofline']['avg_ms'],r['box'])
for k in ('configs2_per_gpu','fp16_mode','configs4_per_gpu','parity_mode','inference'):
    print(k, {a:b for a,b in r[k].items() if a in ('ms_per_step','value','patches_per_s')})
"
