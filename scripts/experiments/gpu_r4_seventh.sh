#!/bin/bash
# round 4, seventh GPU job: walk-ordered pools spread over the launches — the Youtube-like shape, the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1200 python -m pytest tests/test_solver_gpu.py -q -m gpu -k "youtube_scale" -s > $O/tube7.log 2>&1
grep -h "^tube" $O/tube7.log; tail -3 $O/tube7.log
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_solver_gpu.py::test_walk_models_at_youtube_scale_match_the_reference_training_loop > $O/gpu_suite7.log 2>&1
tail -8 $O/gpu_suite7.log
