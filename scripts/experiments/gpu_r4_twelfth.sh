#!/bin/bash
# round 4, twelfth GPU job: hub rows = expected hits >= 1; small tables by chains instead of runs?; the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
E=scripts/experiments/auc_shapes.py
{
timeout 400 python $E hub100k 200 auto 17,18,19 partitions=4 episode=9 hub=auto 2>&1 | grep -E "mean|Error"
timeout 400 python $E hub100k 200 auto 17,18,19 partitions=4 episode=9 2>&1 | grep -E "mean|Error"
timeout 400 python $E hub100k 200 auto 17,18,19 partitions=16 episode=2 hub=auto 2>&1 | grep -E "mean|Error"
timeout 400 python $E blog 2000 auto 17,18,19 hub=auto 2>&1 | grep -E "mean|Error"
timeout 400 python $E hub100k 200 auto 17,18,19 2>&1 | grep -E "mean|Error"
} > $O/small_tables12.log 2>&1
cat $O/small_tables12.log
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite12.log 2>&1
tail -12 $O/gpu_suite12.log
grep -h "^headline\|^tube\|^hub100k" $O/gpu_suite12.log | cut -c1-400
