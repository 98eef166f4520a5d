#!/bin/bash
# one-off: the same at dim 128 (fs128_line_p8), one seed
mkdir -p gpurun_out
timeout 70 python scripts/experiments/configs_auc.py job=fs128_line_p8 seeds=1024 "variants=device=1" 2>&1 | grep -v amdgpu.ids | tail -n 3 > gpurun_out/r5_fs128_device_by_walk_index.log
cat gpurun_out/r5_fs128_device_by_walk_index.log
