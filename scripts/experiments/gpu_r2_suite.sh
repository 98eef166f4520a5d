set -x
python -m pytest tests -m gpu -x -q -k "not hub_heavy and not quick_start_pipeline" 2>&1 | tail -8
python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err; tail -c 3000 gpurun_out/r2_bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end | cut -c1-1200
