#!/bin/bash
# Round 3: hub rows trained by chains (gvk_train_episode_hot).  gpurun --timeout 1500 -- 'bash scripts/experiments/gpu_r3_hot.sh'
mkdir -p gpurun_out
O=gpurun_out/r3_chains.txt
: > $O
timeout 300 python scripts/experiments/hub_debug.py 2>&1 | grep -v amdgpu.ids >> $O
C="variant=0;hub=auto;hub=2048;hub=8192;hub=auto,chain_cap=64;hub=auto,chain_cap=128;hub=auto,chain_cap=1024"
timeout 900 python scripts/experiments/c2_auc.py "configs=$C" 2>&1 | grep -E "^C2|Error|error" >> $O
S=scripts/experiments/auc_shapes.py
for extra in "hub=auto" "hub=auto chain_cap=64" "hub=auto partitions=4 episode=9" "hub=auto partitions=16 episode=2"; do
  timeout 300 python $S hub100k 200 sampled 17,18 $extra 2>&1 | grep -E "mean|Error" >> $O
done
timeout 300 python $S blog 2000 sampled 17,18 hub=auto 2>&1 | grep -E "mean|Error" >> $O
timeout 300 python $S blog 2000 sampled 17,18 hub=auto chain_cap=64 2>&1 | grep -E "mean|Error" >> $O
for extra in "model=DeepWalk aug=5 device_sampling=1" "model=DeepWalk aug=5" "model=node2vec aug=5 p=4 q=2" "model=node2vec aug=5 p=0.25 q=0.25 device_sampling=1"; do
  timeout 300 python $S blog 2000 sampled 17,18 $extra hub=auto 2>&1 | grep -E "mean|Error" >> $O
done
Q="--no-cpu-baseline --no-end-to-end --no-access-pattern --steps 400 --warmup 50"
for T in "" "--hub-rows auto" "--hub-rows auto --tune 8=128" "--hub-rows auto --tune 8=64" "--hub-rows auto --tune 8=32" "--hub-rows 2048 --tune 8=64"; do
  timeout 200 python bench.py $Q $T 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('bench [$T]: %.1f M/s, %.2f us/step, kernel %s %.2f us' % (d['value'], d['ms_per_step'] * 1e3, r['kernel'], r['kernel_ms'] * 1e3))
" >> $O 2>&1
done
cat $O | cut -c1-330
