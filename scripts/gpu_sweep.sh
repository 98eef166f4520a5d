#!/bin/bash
# Dimension sweep of the SGD kernel on the benchmark graph + the full GPU test suite.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "AUC|passed|failed|^E  |^FAILED" > gpurun_out/pytest_gpu6.log
cat gpurun_out/pytest_gpu6.log
for D in 32 64 96 128 256 512; do
  python bench.py --steps 200 --warmup 20 --dim $D --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('dim $D', round(r['value'],1), 'M/s  kernel_ms', round(r['roofline']['kernel_ms'],5), ' algorithmic GB/s', round(r['roofline']['achieved'],1), ' frac', round(r['roofline']['frac'],4))
" | tee -a gpurun_out/dim_sweep.log
done
python bench.py --steps 200 --warmup 20 --negatives 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('dim 128 k=5', round(r['value'],1), 'M/s  kernel_ms', round(r['roofline']['kernel_ms'],5), ' algorithmic GB/s', round(r['roofline']['achieved'],1), ' frac', round(r['roofline']['frac'],4))
" | tee -a gpurun_out/dim_sweep.log
