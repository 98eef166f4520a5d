#!/bin/bash
# Round 6: the judged measurements on one MI355X in one job (outputs under gpurun_out/, summarised into profiles/r6 by
# `python scripts/summarize_profiles.py r6`): GPU suite first (parity is the gate), the driver's bench command, its kernel trace,
# its PMC passes — and then the bench command once more in the SAME job with GVK_BENCH_PMC_SUMMARY pointing at the summary those
# passes produced, so that its `roofline.traffic` is a number of this job and box (bench.py prints null otherwise).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
# REFRESH=1: a later job of the same round on a later tree (the full suite and configs[2..4] are in the first job's files): a subset of
# the suite into pytest_gpu_refresh.log / parity_auc_refresh.log, then the bench command, its trace and its PMC passes as below
LOG=pytest_gpu_full.log; PARITY=parity_auc.log; WHAT="tests -m gpu"
if [ "$REFRESH" = "1" ]; then LOG=pytest_gpu_refresh.log; PARITY=parity_auc_refresh.log; WHAT="tests/test_configs_gpu.py tests/test_hub_chains_gpu.py tests/test_kernel_gpu.py -m gpu"; fi
timeout 2800 python -m pytest $WHAT -q -rPx > $O/$LOG 2>&1
grep -E "passed|failed" $O/$LOG | tail -2; grep -E "^FAILED" $O/$LOG
grep -hE "^(headline|tube|hub100k|blog|module|friendster|youtube|held|hub rows after|a head row|DeepWalk over)|AUC here|hub rows after" $O/$LOG | grep -v "print(" > $O/$PARITY
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_n1_steps20.json 2> $O/bench_n1_steps20.err
tail -c 1500 $O/bench_n1_steps20.json
SHORT="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-module"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernel -- $SHORT > $O/prof_kernel.log 2>&1
for counter in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $counter --output-format csv -d $O/pmc_${counter}_128 -- $SHORT > $O/pmc_${counter}_128.log 2>&1
done
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_L2_128 -- $SHORT > $O/pmc_L2_128.log 2>&1
# traffic by role: the serialized form (--tune 9=1) launches a unit's chains and its pairs one after the other — which of partner rows,
# mirror rows and records the bytes beyond the algorithmic ones are (summarize_profiles.py splits the dispatches by grid size)
for counter in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $counter --output-format csv -d $O/pmc_${counter}_roles -- $SHORT --tune 9=1 > $O/pmc_${counter}_roles.log 2>&1
done
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_L2_roles -- $SHORT --tune 9=1 > $O/pmc_L2_roles.log 2>&1
# the shard size of an 8-GPU run on this one GPU: kernel trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernel_p8 -- $SHORT --partitions 8 > $O/prof_kernel_p8.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -size +30M -delete
# the summary of this job's passes, then the bench line that may quote it
mkdir -p profiles/r6
timeout 300 python scripts/summarize_profiles.py r6 > $O/summarize.log 2>&1; tail -3 $O/summarize.log
if [ -f profiles/r6/pmc_summary_bench_n1.json ]; then
  cp profiles/r6/pmc_summary_bench_n1.json $O/pmc_summary_bench_n1.json
  GVK_BENCH_PMC_SUMMARY=$O/pmc_summary_bench_n1.json timeout 600 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-module > $O/bench_n1_steps20_with_traffic.json 2>> $O/bench_n1_steps20.err
  tail -c 600 $O/bench_n1_steps20_with_traffic.json
fi
timeout 600 python bench.py --steps 400 --warmup 50 --no-end-to-end --no-module > $O/bench_n1.json 2> $O/bench_n1.err
for p in 2 4 8; do timeout 300 python bench.py --steps 200 --warmup 20 --partitions $p --no-cpu-baseline --no-end-to-end --no-module --no-access-pattern 2>/dev/null | tail -n 1; done > $O/bench_by_partitions.jsonl
if [ "$REFRESH" != "1" ]; then timeout 900 python scripts/measure_configs.py --epochs 100 > $O/configs_2_4.jsonl 2> $O/configs_2_4.err; tail -c 400 $O/configs_2_4.jsonl; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
