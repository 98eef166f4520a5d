#!/bin/bash
# round 4, third GPU job: the wide pair body — rate per (wide, parts, lerp), kernel trace, AUC (default, P = 4), the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests3.log 2>&1
tail -5 $O/chains_tests3.log
B="python bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub3.jsonl
for cfg in "8 14 0 3" "8 14 0 2" "8 14 0 1" "8 16 0 3" "8 14 1 2" "6 14 1 2" "5 14 1 2" "5 20 1 2" "10 14 0 3" "4 14 1 2" "1 14 0 3"; do
  set -- $cfg
  echo "parts=$1 cap=$2 lerp=$3 wide=$4" >> $O/bench_hub3.jsonl
  timeout 200 $B --hub-parts $1 --hub-cap $2 --hub-lerp $3 --tune 9=$4 >> $O/bench_hub3.jsonl 2>> $O/bench_hub3.err
done
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub3.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"], j["roofline"].get("kernel_ms"), j["roofline"].get("kernel")[:60])
    elif line:
        print(line)
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_hub8c -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module > $GRAFT_REPO_ROOT/$O/prof_hub8c.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3, collections
for path in glob.glob("gpurun_out/r4/prof_hub8c/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    agg = collections.defaultdict(list)
    for n, s, e in rows:
        agg[n[:60]].append(e - s)
    for n, v in agg.items():
        print(n, len(v), "avg %.2f us" % (sum(v) / len(v) / 1000), "min %.2f max %.2f" % (min(v) / 1000, max(v) / 1000))
PY
timeout 1200 python scripts/experiments/c2_hub.py configs="hub=default;hub=default,lerp=1,parts=6;hub=default,lerp=1,parts=5;hub=default,partitions=4;hub=default,partitions=4,fidelity=throughput" > $O/c2_hub3.log 2>&1
grep "^C2" $O/c2_hub3.log
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_suite3.log 2>&1
tail -25 $O/gpu_suite3.log
