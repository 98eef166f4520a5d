#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r3_chains2.txt
: > $O
timeout 300 python scripts/experiments/hot_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $O
S=scripts/experiments/auc_shapes.py
for extra in "hub=512 partitions=2 episode=9" "hub=512 partitions=4 episode=9" "hub=auto partitions=4 episode=9" "hub=4096 partitions=4 episode=9"; do
  timeout 300 python $S hub100k 20 sampled 17 $extra 2>&1 | grep -E "mean|Error" >> $O
done
C="hub=auto;hub=auto,chain_cap=64;hub=auto,chain_cap=128;hub=auto,chain_cap=1024;hub=1024,chain_cap=128;hub=8192,chain_cap=128"
timeout 900 python scripts/experiments/c2_auc.py "configs=$C" 2>&1 | grep -E "^C2|Error|error" >> $O
for extra in "hub=auto" "hub=auto chain_cap=64" "hub=auto chain_cap=128" "hub=auto chain_cap=1024"; do
  timeout 300 python $S hub100k 200 sampled 17,18 $extra 2>&1 | grep -E "mean|Error" >> $O
done
cat $O | cut -c1-330
