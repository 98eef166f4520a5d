#!/bin/bash
# GPU job: solver e2e tests. Run: gpurun --timeout 1500 -- 'bash scripts/gpu_job.sh'
mkdir -p gpurun_out
python -m pytest tests/test_solver_gpu.py -m gpu -q -s 2>&1 | grep -E "AUC|relative|passed|failed|^E  |^FAILED" > gpurun_out/pytest_gpu5_solver.log
cat gpurun_out/pytest_gpu5_solver.log
python -m pytest tests/test_kernel_gpu.py -m gpu -q 2>&1 | tail -3
