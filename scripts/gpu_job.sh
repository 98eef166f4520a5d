#!/bin/bash
# The round's GPU job: GPU test suite, smoke, bench (default and the driver's short form), BASELINE configs[2..4],
# rocprofv3 kernel trace + PMC passes (per dim) + a marker trace of an end-to-end run.  Run through
#   gpurun --timeout 2400 -- 'bash scripts/gpu_job.sh [tests|measure|profile ...]'
# (no argument = everything).  Summaries are copied into profiles/<round>/ afterwards: scripts/summarize_profiles.py.
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-$PWD}
WHAT=${*:-tests measure profile}
cd $REPO
if [[ $WHAT == *tests* ]]; then
  python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/pytest_gpu_full.log
  # the AUC tables the parity tests print (hub-heavy shapes against the reference's training loop, T3 protocol)
  python -m pytest tests/test_solver_gpu.py -q -s -k "hub_heavy or reference_training_loop or parity" 2>&1 | grep -E "reference loop|AUC|passed|failed" > gpurun_out/parity_auc.log
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
fi
if [[ $WHAT == *measure* ]]; then
  python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
  python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1_steps20.json 2>> gpurun_out/bench_n1.err
  python scripts/measure_configs.py > gpurun_out/configs_2_4.jsonl 2> gpurun_out/configs_2_4.err
  python scripts/measure_engine.py --epochs 1000 > gpurun_out/engine_e2e.jsonl 2> gpurun_out/engine_e2e.err
  for d in 32 64 96 256 512; do
    python bench.py --dim $d --no-cpu-baseline --no-end-to-end --steps 1000 --warmup 100
  done > gpurun_out/dim_sweep.jsonl 2> gpurun_out/dim_sweep.err
  # the same lines carry roofline.access_pattern (gvk_probe_row_traffic): the ceiling of the memory system per dim
  cp gpurun_out/dim_sweep.jsonl gpurun_out/access_pattern_probe.jsonl
  # the shard sizes of multi-GPU runs on the one GPU (what a GPU of an 8-GPU run trains: 16 partitions of 32 MB)
  for parts in 4 8 16; do
    python bench.py --partitions $parts --no-cpu-baseline --no-end-to-end --steps 1000 --warmup 100
  done > gpurun_out/shard_sweep.jsonl 2> gpurun_out/shard_sweep.err
fi
if [[ $WHAT == *profile* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $REPO/gpurun_out/prof_kernel $REPO/gpurun_out/pmc_* $REPO/gpurun_out/prof_marker $REPO/gpurun_out/prof_marker_engine
  Q="--no-cpu-baseline --no-end-to-end"
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_kernel -o trace -- python $REPO/bench.py --steps 200 --warmup 20 $Q > $REPO/gpurun_out/prof_kernel.log 2>&1
  for d in 32 64 96 128 256 512; do
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/pmc_${C}_$d -o pmc -- python $REPO/bench.py --dim $d --steps 20 --warmup 5 $Q > $REPO/gpurun_out/pmc_${C}_$d.log 2>&1
    done
  done
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/gpurun_out/pmc_L2_128 -o pmc -- python $REPO/bench.py --steps 20 --warmup 5 $Q > $REPO/gpurun_out/pmc_L2_128.log 2>&1
  # roctx ranges ("Sample threads", "Upload", "Regroup", "Train Batch", "Exchange") next to the kernels of an end-to-end run
  rocprofv3 --marker-trace --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_marker -o e2e -- python $REPO/scripts/quick_start.py > $REPO/gpurun_out/prof_marker.log 2>&1
  # the same ranges from the native engine (C++ host loop) through the module, CPU samplers then device sampling
  rocprofv3 --marker-trace --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_marker_engine -o engine -- python $REPO/scripts/measure_engine.py --quick --epochs 300 > $REPO/gpurun_out/prof_marker_engine.log 2>&1
  cd $REPO
fi
tail -4 gpurun_out/pytest_gpu_full.log 2>/dev/null; tail -1 gpurun_out/smoke.log 2>/dev/null; tail -c 1200 gpurun_out/bench_n1.json 2>/dev/null
