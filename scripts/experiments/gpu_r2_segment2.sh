set -x
python -m pytest tests/test_kernel_gpu.py -x -q 2>&1 | tail -5
E=scripts/experiments/auc_shapes.py
for shape_ep in "blog 2000" "hub100k 200"; do
  python $E $shape_ep sampled,grouped 17,18,19 steps=1
  python $E $shape_ep sampled,grouped 17,18,19 steps=1 sum=1
done 2>&1 | grep -E "mean|Error|error"
B='python bench.py --no-cpu-baseline --no-end-to-end'
P='import json,sys; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r["value"]), round(r["roofline"]["kernel_ms"]*1e3,2), round(r["roofline"]["frac"],3), r["roofline"]["kernel"], r.get("regroup"))'
for o in sampled grouped; do
  $B --pair-order $o --variant 2 | python -c "$P" "v2 $o"
  $B --pair-order $o --segment-steps 1 | python -c "$P" "seg1 $o"
  $B --pair-order $o --segment-steps 1 --tune 7=0 | python -c "$P" "seg1 with-loss $o"
  $B --pair-order $o --segment-steps 1 --tune 6=1 | python -c "$P" "seg1 sum $o"
  $B --pair-order $o --segment-steps 2 --tune 6=1 | python -c "$P" "seg2 sum $o"
done
for d in 32 64 96; do
  for o in sampled grouped; do
  $B --dim $d --pair-order $o --variant 2 | python -c "$P" "dim $d v2 $o"
  $B --dim $d --pair-order $o --segment-steps 1 | python -c "$P" "dim $d seg1 $o"
  $B --dim $d --pair-order $o --segment-steps 1 --tune 6=1 | python -c "$P" "dim $d seg1 sum $o"
  $B --dim $d --pair-order $o --segment-steps 2 --tune 6=1 | python -c "$P" "dim $d seg2 sum $o"
  done
done
python bench.py --steps 100 --warmup 10 2>&1 | tail -1
