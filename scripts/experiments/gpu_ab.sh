#!/bin/bash
# A/B helper for one-minute GPU jobs: every argument is one bench.py variant ("label|extra bench args|ENV=.. ENV=.."); prints one
# line per variant (rate, launch time x launches, fraction of the HBM peak).  Usage: gpu_ab.sh "base|" "two|--tune 9=2" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
STEPS=${STEPS:-200}
B="python bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-end-to-end --no-module --no-access-pattern"
for v in "$@"; do
  label=${v%%|*}; rest=${v#*|}; extra=${rest%%|*}; envs=${rest#*|}; [ "$envs" = "$rest" ] && envs=""
  env $envs timeout 300 $B $extra > $O/ab_$label.json 2> $O/ab_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$O/ab_$label.json").read().strip().splitlines()[-1])
    f = r["roofline"]
    print("%-22s %8.1f M/s  %7.2f us/batch  kernel %6.2f us x %d  frac %.3f" % ("$label", r["value"], r["ms_per_step"] * 1e3, f["kernel_ms"] * 1e3, f.get("launches_per_step", 1), f["frac"]))
except Exception as e:
    print("%-22s failed: %s" % ("$label", e)); print(open("$O/ab_$label.err").read()[-600:])
PY
done
