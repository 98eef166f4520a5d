#!/bin/bash
# round 4, ninth GPU job: which blocks of the hot kernel's launch come first; parts per batch at P = 4 / 8 (rate and AUC)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests9.log 2>&1
tail -3 $O/chains_tests9.log
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub9.jsonl
for cfg in "0 0" "0 1" "0 2" "0 0" "0 1" "0 2" "5 1" "10 1" "4 1"; do
  set -- $cfg
  echo "parts=$1 order=$2" >> $O/bench_hub9.jsonl
  timeout 200 $B --hub-parts $1 --tune 10=$2 >> $O/bench_hub9.jsonl 2>> $O/bench_hub9.err
done
for cfg in "4 0" "4 16" "4 8" "8 0" "8 16"; do
  set -- $cfg
  echo "partitions=$1 parts=$2 order=1" >> $O/bench_hub9.jsonl
  timeout 200 $B --partitions $1 --hub-parts $2 --tune 10=1 >> $O/bench_hub9.jsonl 2>> $O/bench_hub9.err
done
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub9.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"], j["roofline"].get("kernel_ms"), j["roofline"]["launches_per_step"])
    elif line:
        print(line)
PY
timeout 1500 python scripts/experiments/c2_hub.py configs="hub=default,partitions=4,episode=32,parts=16;hub=default,partitions=4,episode=32,parts=8;hub=default,partitions=8,episode=8,parts=16;hub=default,partitions=4,parts=16;hub=default,partitions=4,parts=8" > $O/c2_hub9.log 2>&1
grep "^C2" $O/c2_hub9.log
