set -x
python -m pytest tests/test_kernel_gpu.py tests/test_bind_gpu.py -x -q 2>&1 | tail -4
python -m pytest tests/test_solver_gpu.py -x -q -k "device_sampled_walks or exchange or several_partitions" 2>&1 | tail -4
B='python bench.py --no-cpu-baseline --no-end-to-end --steps 1000 --warmup 100'
P='import json,sys; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r["value"]), round(r["ms_per_step"]*1e3,2), round(r["roofline"]["kernel_ms"]*1e3,2), round(r["roofline"]["frac"],3), r["roofline"]["kernel"])'
for rep in 1 2 3; do
  for o in sampled grouped; do
    $B --pair-order $o --variant 2 | python -c "$P" "rep$rep v2 $o"
    $B --pair-order $o --segment-steps 1 | python -c "$P" "rep$rep seg1 $o"
  done
done
