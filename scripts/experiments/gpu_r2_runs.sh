set -x
python -m pytest tests/test_kernel_gpu.py -x -q 2>&1 | tail -5
E=scripts/experiments/auc_shapes.py
for shape_ep in "blog 2000" "hub100k 200"; do
  python $E $shape_ep sampled,grouped 17,18,19 variant=2
  python $E $shape_ep sampled,grouped 17,18,19
  python $E $shape_ep grouped 17,18,19 run_cap=4
  python $E $shape_ep grouped 17,18,19 run_cap=64
  python $E $shape_ep sampled,grouped 17,18,19 generation=5120
done 2>&1 | grep -E "mean|Error|error"
for o in sampled grouped; do
  python bench.py --no-cpu-baseline --pair-order $o --variant 2 | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('v2', '$o', r['value'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
  python bench.py --no-cpu-baseline --pair-order $o | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('runs auto', '$o', r['value'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
done
for rc in 1 4 8 64; do
  python bench.py --no-cpu-baseline --pair-order grouped --run-cap $rc | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('runs cap $rc grouped', r['value'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
done
