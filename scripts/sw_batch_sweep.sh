#!/bin/bash
# configs[3] inference leg by windows per launch: bash scripts/sw_batch_sweep.sh "4 6 9 3"
for b in $1; do
  python3 bench.py --steps 3 --warmup 1 --sw-batch $b --infer-headline-only --no-cpu-baseline --no-parity-leg --no-other-configs 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('sw_batch', $b, 'inference', r['inference']['value'], 'x3', r['inference_x3']['value'])"
done
