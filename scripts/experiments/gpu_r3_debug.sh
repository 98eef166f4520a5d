#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r3_debug.txt
: > $O
timeout 600 python scripts/experiments/hot_check.py >> $O 2>&1
timeout 600 python scripts/experiments/hub_debug.py 200000 2000000 2000 1 >> $O 2>&1
timeout 600 python scripts/experiments/hub_debug.py 1000000 10000000 100000 3 >> $O 2>&1
Q="--no-cpu-baseline --no-end-to-end --no-access-pattern --steps 400 --warmup 50"
for T in "" "--hub-rows auto" "--hub-rows auto --tune 8=128" "--hub-rows auto --tune 8=64"; do
  timeout 200 python bench.py $Q $T 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('bench [$T]: %.1f M/s, %.2f us/step, kernel %s %.2f us' % (d['value'], d['ms_per_step'] * 1e3, r['kernel'], r['kernel_ms'] * 1e3))
" >> $O 2>&1
done
cat $O | cut -c1-400
