#!/bin/bash
# round 4, sixth GPU job: the hot kernel at four wavefronts per SIMD — kernel tests, rate, chains and pairs apart, AUC, the GPU suite (with the Youtube-like walk shape)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests6.log 2>&1
tail -5 $O/chains_tests6.log
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub6.jsonl
for cfg in "0 0 -1" "8 7 1" "5 7 1" "10 7 0" "4 7 1" "20 7 0"; do
  set -- $cfg
  echo "parts=$1 cap=$2 lerp=$3" >> $O/bench_hub6.jsonl
  timeout 200 $B --hub-parts $1 --hub-cap $2 --hub-lerp $3 >> $O/bench_hub6.jsonl 2>> $O/bench_hub6.err
done
echo "partitions=4" >> $O/bench_hub6.jsonl
timeout 200 $B --partitions 4 >> $O/bench_hub6.jsonl 2>> $O/bench_hub6.err
echo "fidelity=throughput" >> $O/bench_hub6.jsonl
timeout 200 $B --fidelity throughput >> $O/bench_hub6.jsonl 2>> $O/bench_hub6.err
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub6.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"], j["roofline"].get("kernel_ms"), j["roofline"].get("kernel")[:70])
    elif line:
        print(line)
PY
for mode in 1 0; do
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof6_$mode -- $B --tune 9=$mode > $GRAFT_REPO_ROOT/$O/prof6_$mode.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3, collections, statistics
for mode in (1, 0):
  for path in glob.glob("gpurun_out/r4/prof6_%d/*/*_results.db" % mode):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
    agg = collections.defaultdict(list)
    for n, s, e, g in rows:
        agg[(n[:50], g)].append(e - s)
    print("serialized" if mode else "fused")
    for (n, g), v in sorted(agg.items(), key=lambda x: -len(x[1]))[:3]:
        print(" ", n, "grid", g, len(v), "avg %.2f us" % (sum(v) / len(v) / 1000), "median %.2f min %.2f max %.2f" % (statistics.median(v) / 1000, min(v) / 1000, max(v) / 1000))
PY
timeout 1500 python scripts/experiments/c2_hub.py configs="hub=default;hub=default,lerp=1;hub=default,partitions=8,episode=8;hub=default,partitions=2,episode=128" > $O/c2_hub6.log 2>&1
grep "^C2" $O/c2_hub6.log
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite6.log 2>&1
tail -12 $O/gpu_suite6.log
grep -h "^tube\|^blog\|^hub100k\|^headline" $O/gpu_suite6.log | head -40
