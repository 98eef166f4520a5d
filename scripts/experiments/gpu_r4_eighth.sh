#!/bin/bash
# round 4, eighth GPU job: pair steps per wavefront, long tasks with one wait — kernel tests, rate, chains and pairs apart
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py tests/test_kernel_gpu.py -x -q -m gpu -k "chains or parts or hub or spread" > $O/chains_tests8.log 2>&1
tail -3 $O/chains_tests8.log
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub8.jsonl
for cfg in "0 1" "0 2" "0 3" "0 4" "0 8" "5 2" "5 4" "10 2" "4 4"; do
  set -- $cfg
  echo "parts=$1 pair_steps=$2" >> $O/bench_hub8.jsonl
  timeout 200 $B --hub-parts $1 --tune 10=$2 >> $O/bench_hub8.jsonl 2>> $O/bench_hub8.err
done
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub8.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"], j["roofline"].get("kernel_ms"))
    elif line:
        print(line)
PY
for mode in "1 1" "1 2" "1 4" "0 2"; do
set -- $mode
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof8_$1_$2 -- $B --tune 9=$1 --tune 10=$2 > $GRAFT_REPO_ROOT/$O/prof8_$1_$2.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3, collections, statistics
for path in sorted(glob.glob("gpurun_out/r4/prof8_*/*/*_results.db")):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
    agg = collections.defaultdict(list)
    for n, s, e, g in rows:
        agg[(n[:50], g)].append(e - s)
    print(path.split("/")[2])
    for (n, g), v in sorted(agg.items(), key=lambda x: -len(x[1]))[:2]:
        print(" ", n, "grid", g, len(v), "avg %.2f us" % (sum(v) / len(v) / 1000), "median %.2f min %.2f max %.2f" % (statistics.median(v) / 1000, min(v) / 1000, max(v) / 1000))
PY
