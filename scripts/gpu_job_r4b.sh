#!/bin/bash
# Round 4, second session: the judged measurements of the final tree on one MI355X (outputs under gpurun_out/, copied into
# profiles/r4 by scripts/summarize_profiles.py r4): the driver's bench command, its kernel trace, smoke, the GPU suite (last: the round's
# GPU minutes end with it).  The PMC passes of scripts/gpu_job_r4.sh are not repeated: train_hot_kernel is the kernel they measured.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_n1_steps20.json 2> $O/bench_n1_steps20.err
tail -c 1200 $O/bench_n1_steps20.json
SHORT="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-module"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernel -- $SHORT > $O/prof_kernel.log 2>&1
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -n 1 $O/smoke.log
timeout 2800 python -m pytest tests -v -m gpu -rP --durations=15 > $O/pytest_gpu_full.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_full.log | tail -n 2; grep -E "^FAILED" $O/pytest_gpu_full.log
grep -hE "^(headline|tube|hub100k|blog|AUC here|module)" $O/pytest_gpu_full.log > $O/parity_auc.log
find $O -name "*kernel_trace.csv" -size +30M -delete
