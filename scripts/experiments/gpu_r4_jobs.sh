#!/bin/bash
# The round-4 GPU jobs behind profiles/r4/experiments/r4_rate_by_configuration.txt, r4_c2_auc.txt, r4_chains_and_pairs_apart.txt
# and r4_small_tables.txt, one after the other as the executor changed (each ran through gpurun on one MI355X):
#     bash scripts/experiments/gpu_r4_jobs.sh <first | second | ... | thirteenth>
# Some of their knobs (bench.py --tune keys of bring-up forms) no longer exist in the final tree: the lines they produced do.
job=${1:?which job: first .. thirteenth}
case "$job" in
first)
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
;;
second)
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
;;
third)
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
;;
fourth)
# round 4, fourth GPU job: chains and pairs apart in a kernel trace; AUC of the default on the headline shape at P = 1, 2, 4; the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_serialized -- $B --tune 9=1 > $GRAFT_REPO_ROOT/$O/prof_serialized.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3, collections, statistics
for path in glob.glob("gpurun_out/r4/prof_serialized/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
    agg = collections.defaultdict(list)
    for n, s, e, g in rows:
        agg[(n[:50], g)].append(e - s)
    for (n, g), v in sorted(agg.items(), key=lambda x: -len(x[1]))[:8]:
        print(n, "grid", g, len(v), "avg %.2f us" % (sum(v) / len(v) / 1000), "median %.2f min %.2f max %.2f" % (statistics.median(v) / 1000, min(v) / 1000, max(v) / 1000))
PY
timeout 1500 python scripts/experiments/c2_hub.py configs="hub=default;hub=default,lerp=1;hub=default,partitions=4;hub=default,partitions=4,parts=8;hub=default,partitions=2;hub=default,device=1" > $O/c2_hub4.log 2>&1
grep "^C2" $O/c2_hub4.log
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_suite4.log 2>&1
tail -30 $O/gpu_suite4.log
;;
fifth)
# round 4, fifth GPU job: tasks of seven entries with records — kernel tests, rate, chains and pairs apart, AUC (P = 1, 2, 4), the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 900 python -m pytest tests/test_hub_chains_gpu.py -x -q -m gpu > $O/chains_tests5.log 2>&1
tail -5 $O/chains_tests5.log
B="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-end-to-end --no-cpu-baseline --no-access-pattern --no-module"
: > $O/bench_hub5.jsonl
for cfg in "0 0 0" "8 7 0" "8 4 0" "8 7 1" "5 7 1" "10 7 0" "4 7 1"; do
  set -- $cfg
  echo "parts=$1 cap=$2 lerp=$3" >> $O/bench_hub5.jsonl
  timeout 200 $B --hub-parts $1 --hub-cap $2 --hub-lerp $3 >> $O/bench_hub5.jsonl 2>> $O/bench_hub5.err
done
python - <<'PY'
import json
for line in open("gpurun_out/r4/bench_hub5.jsonl"):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print("%.1f M/s" % j["value"], "%.2f us/step" % (1000 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"], j["roofline"].get("kernel_ms"), j["roofline"].get("kernel")[:70])
    elif line:
        print(line)
PY
for mode in 1 0; do
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof5_$mode -- $B --tune 9=$mode > $GRAFT_REPO_ROOT/$O/prof5_$mode.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3, collections, statistics
for mode in (1, 0):
  for path in glob.glob("gpurun_out/r4/prof5_%d/*/*_results.db" % mode):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
    agg = collections.defaultdict(list)
    for n, s, e, g in rows:
        agg[(n[:50], g)].append(e - s)
    print("serialized" if mode else "fused")
    for (n, g), v in sorted(agg.items(), key=lambda x: -len(x[1]))[:4]:
        print(" ", n, "grid", g, len(v), "avg %.2f us" % (sum(v) / len(v) / 1000), "median %.2f min %.2f max %.2f" % (statistics.median(v) / 1000, min(v) / 1000, max(v) / 1000))
PY
timeout 1500 python scripts/experiments/c2_hub.py configs="hub=default;hub=default,lerp=1;hub=default,parts=5,lerp=1;hub=default,partitions=4;hub=default,partitions=2;hub=default,partitions=4,episode=32" > $O/c2_hub5.log 2>&1
grep "^C2" $O/c2_hub5.log
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_suite5.log 2>&1
tail -12 $O/gpu_suite5.log
;;
sixth)
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
;;
seventh)
# round 4, seventh GPU job: walk-ordered pools spread over the launches — the Youtube-like shape, the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1200 python -m pytest tests/test_solver_gpu.py -q -m gpu -k "youtube_scale" -s > $O/tube7.log 2>&1
grep -h "^tube" $O/tube7.log; tail -3 $O/tube7.log
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_solver_gpu.py::test_walk_models_at_youtube_scale_match_the_reference_training_loop > $O/gpu_suite7.log 2>&1
tail -8 $O/gpu_suite7.log
;;
eighth)
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
;;
ninth)
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
;;
tenth)
# round 4, tenth GPU job: the margin of the default on the headline shape — hub rows, parts, lerp, two seeds each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1700 python scripts/experiments/c2_hub.py seeds=1024,5,6 configs="hub=default;hub=default,lerp=1;hub=6000;hub=12000;hub=default,parts=10;hub=default,parts=10,lerp=1;hub=6000,lerp=1;hub=6000,parts=10,lerp=1" > $O/c2_hub10.log 2>&1
grep "^C2" $O/c2_hub10.log
;;
eleventh)
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
;;
twelfth)
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
;;
thirteenth)
# round 4, thirteenth GPU job: the GPU suite after the rule change (chains for cache-resident tables where feasible), parity lines kept
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 2800 python -m pytest tests -q -m gpu -rP > $O/gpu_suite13.log 2>&1
grep -E "passed|failed" $O/gpu_suite13.log | tail -3
grep -E "^FAILED" $O/gpu_suite13.log
grep -hE "^(headline|tube|hub100k|blog|AUC here|module)" $O/gpu_suite13.log | cut -c1-330
;;
*) echo "unknown job $job" >&2; exit 2 ;;
esac
