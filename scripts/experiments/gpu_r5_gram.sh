#!/bin/bash
# Round 5, first GPU job (about two minutes of box time): the Gram-matrix form of the long chains (GVK_TUNE_HOT_GRAM = 11,
# long_chain_gram in graphvite_amd/csrc/gvk_kernels.hip), which was written in round 4 after the GPU minutes had run out.
#   1. its parity test against the oracle (cap 16, 64 tasks) — if this fails, stop: tests/test_gram_chain_cpu.py is the twin to
#      compare the device against, statement by statement;
#   2. A/B on the headline shape and at the shard size of an 8-GPU run, same box, interleaved;
#   3. AUC on the headline shape with it on (scripts/experiments/c2_hub.py tune11=1) next to the default.
# Usage: gpurun --timeout 600 -- 'bash scripts/experiments/gpu_r5_gram.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
GVK_TEST_GRAM=1 timeout 240 python -m pytest tests/test_hub_chains_gpu.py -q -k gram_matrices -x 2>&1 | tail -15 | tee $O/r5_gram_test.log
if ! grep -q "passed" $O/r5_gram_test.log || grep -q "failed" $O/r5_gram_test.log; then echo "gram test red: no timing"; exit 1; fi
STEPS=200 bash scripts/experiments/gpu_ab.sh "steps|" "gram|--tune 11=1" "gram4w|--tune 11=2" "steps2|" "gram2|--tune 11=1" "gram4w2|--tune 11=2" \
  "gram_chains_first|--tune 11=1 --tune 10=0" "p8_steps|--partitions 8" "p8_gram|--partitions 8 --tune 11=1" 2>&1 | tee $O/r5_gram_ab.log
timeout 200 python scripts/experiments/c2_hub.py 'configs=hub=default;hub=default,tune11=1;hub=default,tune11=1,partitions=8,episode=8' 2>&1 | tail -6 | tee $O/r5_gram_auc.log
