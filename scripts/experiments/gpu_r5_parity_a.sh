#!/bin/bash
# Round 5: the chain tests (incl. the batch-level pin against the sequential loop) and the first T3 lines of the new goldens.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_hub_chains_gpu.py -q -x -rP 2>&1 | grep -E "passed|failed|hub rows after|a head row|Error|error|assert" | tee $O/r5_chain_tests.log
timeout 900 python -m pytest tests/test_configs_gpu.py -q -rP -k "${CONFIG_TESTS:-youtube or held}" 2>&1 | grep -E "passed|failed|skipped|AUC here|MARGINAL|OUTSIDE|Error|error|assert|hub_rows" | tee $O/r5_configs_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-module > $O/r5_bench_quick.json 2> $O/r5_bench_quick.err; tail -c 900 $O/r5_bench_quick.json
