#!/bin/bash
# round 4, first GPU job: the mirrors executor — kernel tests, rate per (parts, cap, lerp), AUC on the headline shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests.log 2>&1
tail -5 $O/chains_tests.log
B="python bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
timeout 200 $B > $O/bench_base.json 2> $O/bench_base.err
: > $O/bench_hub.jsonl
for cfg in "8 16 0" "8 16 1" "8 32 0" "5 16 1" "5 32 1" "10 16 0" "10 32 0" "4 32 1" "20 16 0" "1 16 0"; do
  set -- $cfg
  echo "parts=$1 cap=$2 lerp=$3" >> $O/bench_hub.jsonl
  timeout 200 $B --hub-rows auto --hub-parts $1 --hub-cap $2 --hub-lerp $3 >> $O/bench_hub.jsonl 2>> $O/bench_hub.err
done
python - <<'PY'
import json
for name in ("bench_base.json", "bench_hub.jsonl"):
    for line in open("gpurun_out/r4/" + name):
        line = line.strip()
        if line.startswith("{"):
            j = json.loads(line)
            print(name, "%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), j["roofline"].get("kernel"), j["roofline"].get("kernel_ms"))
        elif line:
            print(line)
PY
timeout 1500 python scripts/experiments/c2_hub.py configs="hub=auto,parts=8;hub=auto,parts=8,lerp=1;hub=auto,parts=5,lerp=1,cap=32;hub=auto,parts=10;hub=auto,parts=20,cap=32;hub=0" > $O/c2_hub.log 2>&1
cat $O/c2_hub.log | grep -v "^$" | tail -12
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_hub8 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module --hub-rows auto --hub-parts 8 > $GRAFT_REPO_ROOT/$O/prof_hub8.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof_hub8 -name "*kernel_stats*" | head -1 | xargs -r head -8
find $O/prof_hub8 -name "*_kernel_trace.csv" -size +20M -delete
