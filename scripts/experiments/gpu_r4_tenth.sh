#!/bin/bash
# round 4, tenth GPU job: the margin of the default on the headline shape — hub rows, parts, lerp, two seeds each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1700 python scripts/experiments/c2_hub.py seeds=1024,5,6 configs="hub=default;hub=default,lerp=1;hub=6000;hub=12000;hub=default,parts=10;hub=default,parts=10,lerp=1;hub=6000,lerp=1;hub=6000,parts=10,lerp=1" > $O/c2_hub10.log 2>&1
grep "^C2" $O/c2_hub10.log
