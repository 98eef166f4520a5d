#!/bin/bash
# A/B: embedding rows loaded (1) / loaded and stored (2) with the non-temporal cache hint, so that streaming rows do not
# displace the negative sampler's alias table (8 MB at 1M rows) from the L2s.  Variant libraries are built with
#   hipcc ... -DGVK_EXPERIMENT_NT_ROWS={1,2} -c gvk_kernels.hip   (graphvite_amd/csrc/build/variants/, not shipped)
# Run through gpurun: bash scripts/experiments/gpu_r2_nt.sh > gpurun_out/r2_nt.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
cp graphvite_amd/libgvk.so /tmp/libgvk_shipped.so
Q="--no-cpu-baseline --no-end-to-end --no-access-pattern --steps 1000 --warmup 100"
for round in 1 2; do
  for v in shipped nt1 nt2; do
    if [ $v == shipped ]; then cp /tmp/libgvk_shipped.so graphvite_amd/libgvk.so; else cp graphvite_amd/csrc/build/variants/libgvk_$v.so graphvite_amd/libgvk.so; fi
    for d in 32 64 128; do
      python bench.py --dim $d $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$v dim $d round $round: %.1f M edge-samples/s, kernel %.2f us, %.3f of peak' % (d['value'], r['kernel_ms'] * 1e3, r['frac']))"
    done
  done
done
cp /tmp/libgvk_shipped.so graphvite_amd/libgvk.so
