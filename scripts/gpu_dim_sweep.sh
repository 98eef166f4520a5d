#!/bin/bash
# Dimension / negatives sweep of the SGD kernel on the benchmark graph, default pair order (auto) and sampler order.
mkdir -p gpurun_out
rm -f gpurun_out/dim_sweep.log
for O in auto sampled; do
for A in "--dim 32" "--dim 64" "--dim 96" "--dim 128" "--dim 256" "--dim 512" "--dim 128 --negatives 5"; do
  python bench.py --steps 200 --warmup 20 $A --pair-order $O --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$A', 'pair_order', r['config']['pair_order'][:7], round(r['value'],1), 'M/s  kernel_ms', round(r['roofline']['kernel_ms'],5), ' algorithmic GB/s', round(r['roofline']['achieved'],1), ' frac', round(r['roofline']['frac'],4))
" | tee -a gpurun_out/dim_sweep.log
done
done
