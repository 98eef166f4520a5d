#!/bin/bash
# round 4, second GPU job: work lists with records, copy blocks, per-task label counts — rate per (parts, cap, lerp), kernel trace, then the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests2.log 2>&1
tail -5 $O/chains_tests2.log
B="python bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub2.jsonl
for cfg in "8 14 0" "8 16 0" "8 14 1" "6 14 1" "5 14 1" "10 14 0" "4 14 1" "1 14 0"; do
  set -- $cfg
  echo "parts=$1 cap=$2 lerp=$3" >> $O/bench_hub2.jsonl
  timeout 200 $B --hub-rows auto --hub-parts $1 --hub-cap $2 --hub-lerp $3 >> $O/bench_hub2.jsonl 2>> $O/bench_hub2.err
done
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub2.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), j["roofline"].get("kernel"), j["roofline"].get("kernel_ms"))
    elif line:
        print(line)
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_hub8b -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module --hub-rows auto --hub-parts 8 --hub-cap 14 > $GRAFT_REPO_ROOT/$O/prof_hub8b.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3, statistics, collections
for path in glob.glob("gpurun_out/r4/prof_hub8b/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    agg = collections.defaultdict(list)
    for n, s, e in rows:
        agg[n[:60]].append(e - s)
    for n, v in agg.items():
        print(n, len(v), "avg %.2f us" % (sum(v) / len(v) / 1000), "min %.2f max %.2f" % (min(v) / 1000, max(v) / 1000))
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
tail -15 $O/gpu_suite.log
