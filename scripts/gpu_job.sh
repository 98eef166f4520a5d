#!/bin/bash
# The round's GPU job: GPU test suite, smoke, bench (the driver's short form and the long form), rocprofv3 kernel trace +
# PMC passes of the driver's command, the Friendster-shard kernel (dim 96) with its PMC passes, BASELINE configs[2..4].
#   gpurun --timeout 2400 -- 'bash scripts/gpu_job.sh [tests|measure|profile|configs ...]'      (no argument = everything)
# Summaries are copied into profiles/<round>/ afterwards: python scripts/summarize_profiles.py r3
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-$PWD}
WHAT=${*:-tests measure profile configs}
cd $REPO
Q="--no-cpu-baseline --no-end-to-end"
if [[ $WHAT == *tests* ]]; then
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_gpu_full.log
  # the AUC tables the parity tests print (against the reference's training loop / the sequential pipeline)
  timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_bind_gpu.py -q -s -k "reference_training_loop or parity or sequential or hub_heavy" 2>&1 | grep -E "AUC|passed|failed" > gpurun_out/parity_auc.log
  timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
fi
if [[ $WHAT == *measure* ]]; then
  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1_steps20.json 2> gpurun_out/bench_n1.err
  timeout 900 python bench.py $Q > gpurun_out/bench_n1.json 2>> gpurun_out/bench_n1.err
  for d in 32 64 96 256 512; do
    timeout 900 python bench.py --dim $d $Q --steps 400 --warmup 50
  done > gpurun_out/dim_sweep.jsonl 2> gpurun_out/dim_sweep.err
  cp gpurun_out/dim_sweep.jsonl gpurun_out/access_pattern_probe.jsonl
  # the shard sizes of multi-GPU runs on the one GPU (what a GPU of an N-GPU run trains: P = N partitions)
  for parts in 2 4 8 16; do
    timeout 900 python bench.py --partitions $parts $Q --steps 400 --warmup 50
  done > gpurun_out/shard_sweep.jsonl 2> gpurun_out/shard_sweep.err
fi
if [[ $WHAT == *profile* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $REPO/gpurun_out/prof_kernel $REPO/gpurun_out/pmc_*
  # the driver's own command (bench.py --steps 20 --warmup 5), end-to-end legs off
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_kernel -o trace -- python $REPO/bench.py --steps 20 --warmup 5 $Q > $REPO/gpurun_out/prof_kernel.log 2>&1
  for d in 64 96 128; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/pmc_${C}_$d -o pmc -- python $REPO/bench.py --dim $d --steps 20 --warmup 5 $Q > $REPO/gpurun_out/pmc_${C}_$d.log 2>&1
    done
  done
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_L2_128 -o pmc -- python $REPO/bench.py --steps 20 --warmup 5 $Q > $REPO/gpurun_out/pmc_L2_128.log 2>&1
  # configs[4]'s kernel: dim 96 on one Friendster shard (8.2M rows = 3.1 GB per table), kernel + probe, then its PMC passes
  F="--dim 96 --vertices 8200000 --edges 82000000 --steps 200 --warmup 20 $Q"
  timeout 900 python $REPO/bench.py $F > $REPO/gpurun_out/friendster_shard.json 2> $REPO/gpurun_out/friendster_shard.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/pmc_${C}_shard96 -o pmc -- python $REPO/bench.py $F --no-access-pattern > $REPO/gpurun_out/pmc_${C}_shard96.log 2>&1
  done
  cd $REPO
fi
if [[ $WHAT == *configs* ]]; then
  timeout 900 python scripts/measure_friendster.py > gpurun_out/configs_4.jsonl 2> gpurun_out/configs_4.err
  timeout 900 python scripts/measure_configs.py --skip-friendster > gpurun_out/configs_2_4.jsonl 2> gpurun_out/configs_2_4.err
fi
tail -4 gpurun_out/pytest_gpu_full.log 2>/dev/null; tail -1 gpurun_out/smoke.log 2>/dev/null; tail -c 1500 gpurun_out/bench_n1_steps20.json 2>/dev/null
