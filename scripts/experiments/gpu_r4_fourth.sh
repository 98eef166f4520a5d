#!/bin/bash
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
