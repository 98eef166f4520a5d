#!/bin/bash
# Full GPU job: GPU test suite, smoke, bench, rocprofv3 kernel trace + PMC passes of the same bench command.
# Run through: gpurun --timeout 2400 -- 'bash scripts/gpu_job.sh'. Summaries are copied into profiles/ afterwards
# (scripts/summarize_profiles.py).
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-$PWD}
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/prof_r1 $REPO/gpurun_out/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r1 -o r1 -- python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $REPO/gpurun_out/prof_bench.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/pmc_$C -o pmc -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/pmc_$C.log 2>&1
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_L2 -o pmc -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/pmc_L2.log 2>&1
cd $REPO
tail -4 gpurun_out/pytest_gpu_full.log; tail -1 gpurun_out/smoke.log; tail -c 1500 gpurun_out/bench_n1.json
