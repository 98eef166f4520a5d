#!/bin/bash
# A/B: the negative sampler's table — one alias slot per row (8 MB at 1M rows: a memory request per draw) against an
# alias table over the classes of equal-degree rows (a few thousand 16-byte entries, cache-resident), same distribution.
# Run through gpurun: bash scripts/experiments/gpu_r2_classes.sh > gpurun_out/r2_classes.txt
Q="--no-cpu-baseline --no-end-to-end --steps 1000 --warmup 100"
for round in 1 2; do
  for t in rows classes; do
    for d in 32 64 96 128 256; do
      python bench.py --dim $d --negative-table $t $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; a = r.get('access_pattern') or {}
print('$t dim $d round $round: %.1f M edge-samples/s, kernel %.2f us, %.3f of peak, probe %.2f us, kernel/probe %.3f | %s' % (d['value'], r['kernel_ms'] * 1e3, r['frac'], a.get('kernel_ms', 0) * 1e3, a.get('train_kernel_vs_probe', 0), d['config']['negative_table'][:24]))"
    done
  done
done
