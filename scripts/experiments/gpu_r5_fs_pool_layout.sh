#!/bin/bash
# The jobs behind DESIGN.md section 7.11 a (profiles/r5/experiments/r5_fs_mixed_cpu_pools.txt, r5_fs_shuffle_base.txt,
# r5_fs_device_by_walk_index.txt): fs_line_p8 (Friendster-like 2M / 40M, dim 96, LINE aug 2, 8 partitions) against the
# reference's loop, by WHERE the pairs of one walk lie in a pool.  About 15 s of GPU per training.
#   gpurun --timeout 600 -- 'bash scripts/experiments/gpu_r5_fs_pool_layout.sh'
mkdir -p gpurun_out
out=gpurun_out/r5_fs_pool_layout.log
: > $out
run() { timeout 300 python scripts/experiments/configs_auc.py job=$1 seeds=$2 "variants=$3" 2>&1 | grep -v amdgpu.ids | tail -n 12 >> $out; }
# the CPU samplers' pools under other shuffle bases (the pairs of a walk capacity / base records apart), default executor and pair by pair
run fs_line_p8 1024,5 "sb=40;sb=200;sb=400;sb=1000;sb=4000;sb=100000;sb=1000,fidelity=throughput"
# positives drawn on the device: the reference's base, and larger ones
run fs_line_p8 1024,5 "device=1;device=1,sb=40;device=1,sb=1000"
run fs128_line_p8 1024 "device=1"
timeout 120 python -m pytest tests/test_kernel_gpu.py -q -m gpu -k sample_walks 2>&1 | tail -n 3 >> $out
cat $out
