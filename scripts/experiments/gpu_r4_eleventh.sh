#!/bin/bash
# round 4, eleventh GPU job: rows copied between mirrors only while a mirror is behind — kernel tests; more hub rows: AUC (three seeds) and rate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests11.log 2>&1
tail -3 $O/chains_tests11.log
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub11.jsonl
for hub in default 7840 12534 16384; do
  echo "hub=$hub" >> $O/bench_hub11.jsonl
  timeout 200 $B --hub-rows $hub >> $O/bench_hub11.jsonl 2>> $O/bench_hub11.err
done
echo "hub=16384 lerp" >> $O/bench_hub11.jsonl
timeout 200 $B --hub-rows 16384 --hub-lerp 1 >> $O/bench_hub11.jsonl 2>> $O/bench_hub11.err
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub11.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"], j["roofline"].get("kernel_ms"), j["roofline"]["launches_per_step"])
    elif line:
        print(line)
PY
timeout 1700 python scripts/experiments/c2_hub.py seeds=1024,5,6 configs="hub=default;hub=7840;hub=12534;hub=16384;hub=16384,lerp=1" > $O/c2_hub11.log 2>&1
grep "^C2" $O/c2_hub11.log
