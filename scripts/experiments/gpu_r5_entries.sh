#!/bin/bash
# Round 5: the long chains' entries side by side (entry_sums, gvk_kernels.hip) on the MI355X: the chain tests against the oracle,
# then A/B against round 4's library (graphvite_amd/csrc/build/old/libgvk_r4.so, built from commit 15209b6) on one box, interleaved,
# at P = 1 and at the shard size of an 8-GPU run, the block orders of a launch, the stamps of a launch, and the AUC on the headline shape.
# Usage: gpurun --timeout 900 -- 'bash scripts/experiments/gpu_r5_entries.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 400 python -m pytest tests/test_hub_chains_gpu.py -q -x 2>&1 | tail -25 | tee $O/r5_entries_test.log
OLD="GVK_ALLOW_TEST_LIBRARY=1 GVK_LIBRARY=graphvite_amd/csrc/build/old/libgvk_r4.so"
STEPS=200 bash scripts/experiments/gpu_ab.sh "r4||$OLD" "new|" "new_order0|--tune 10=0" "new_order2|--tune 10=2" "r4_b||$OLD" "new_b|" \
  "p8_r4|--partitions 8|$OLD" "p8_new|--partitions 8" "p8_new_order0|--partitions 8 --tune 10=0" "p4_r4|--partitions 4|$OLD" "p4_new|--partitions 4" 2>&1 | tee $O/r5_entries_ab.log
if [ -f graphvite_amd/csrc/build/ts/libgvk_ts.so ]; then bash scripts/experiments/gpu_r4_stamps.sh 2>&1 | tail -60 | tee $O/r5_entries_stamps.log; fi
timeout 300 python scripts/experiments/c2_hub.py 'configs=hub=default;hub=default,partitions=8,episode=8;hub=default,partitions=4,episode=32' seeds=1024,5 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/r5_entries_auc.log
