# Round 3: the one-orchestrator stack on the GPU — bench.py over engine sessions, the thin Python solver, the module.
set -x
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_engine.json 2> gpurun_out/r3_bench_engine.err
tail -c 600 gpurun_out/r3_bench_engine.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-end-to-end --partitions 16 >> gpurun_out/r3_bench_engine.json 2>> gpurun_out/r3_bench_engine.err
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-end-to-end --partitions 4 >> gpurun_out/r3_bench_engine.json 2>> gpurun_out/r3_bench_engine.err
timeout 1200 python -m pytest tests/test_solver_gpu.py tests/test_bind_gpu.py -q -s -k "not walk_models" 2>&1 | grep -E "AUC|passed|failed|Error|error|assert" > gpurun_out/r3_engine_tests.txt
tail -50 gpurun_out/r3_engine_tests.txt
