#!/bin/bash
# The round's judged measurements on one MI355X (outputs under gpurun_out/, summarised into profiles/r4 by
# scripts/summarize_profiles.py r4): the driver's bench command, its kernel trace, its PMC passes, the GPU suite, smoke.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout 900 python scripts/experiments/c2_hub.py seeds=1024,5 configs="hub=default;hub=default,partitions=4,episode=32;hub=default,partitions=8,episode=8;hub=default,partitions=2,episode=128" > $O/c2_parity.log 2>&1
grep "^C2" $O/c2_parity.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_n1_steps20.json 2> $O/bench_n1_steps20.err
tail -c 1500 $O/bench_n1_steps20.json
SHORT="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-module"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernel -- $SHORT > $O/prof_kernel.log 2>&1
for counter in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $counter --output-format csv -d $O/pmc_${counter}_128 -- $SHORT > $O/pmc_${counter}_128.log 2>&1
done
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_L2_128 -- $SHORT > $O/pmc_L2_128.log 2>&1
cd $R
timeout 600 python bench.py --steps 400 --warmup 50 --no-end-to-end --no-module > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --steps 20 --warmup 5 --fidelity throughput --no-end-to-end --no-module --no-cpu-baseline > $O/bench_n1_steps20_throughput.json 2>> $O/bench_n1.err
timeout 2800 python -m pytest tests -q -m gpu -rP > $O/pytest_gpu_full.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_full.log | tail -2; grep -E "^FAILED" $O/pytest_gpu_full.log
grep -hE "^(headline|tube|hub100k|blog|AUC here|module)" $O/pytest_gpu_full.log > $O/parity_auc.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
find $O -name "*kernel_trace.csv" -size +30M -delete
