#!/bin/bash
# round 4, thirteenth GPU job: the GPU suite after the rule change (chains for cache-resident tables where feasible), parity lines kept
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 2800 python -m pytest tests -q -m gpu -rP > $O/gpu_suite13.log 2>&1
grep -E "passed|failed" $O/gpu_suite13.log | tail -3
grep -E "^FAILED" $O/gpu_suite13.log
grep -hE "^(headline|tube|hub100k|blog|AUC here|module)" $O/gpu_suite13.log | cut -c1-330
