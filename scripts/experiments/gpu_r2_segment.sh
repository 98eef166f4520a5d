set -x
python -m pytest tests/test_kernel_gpu.py -x -q 2>&1 | tail -5
E=scripts/experiments/auc_shapes.py
for shape_ep in "blog 2000" "hub100k 200"; do
  python $E $shape_ep sampled,grouped 17,18,19 variant=2
  python $E $shape_ep sampled,grouped 17,18,19 steps=1
  python $E $shape_ep sampled,grouped 17,18,19 steps=2
  python $E $shape_ep sampled,grouped 17,18,19 steps=4
done 2>&1 | grep -E "mean|Error|error"
B='python bench.py --no-cpu-baseline'
P='import json,sys; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r["value"]), round(r["roofline"]["kernel_ms"]*1e3,2), round(r["roofline"]["frac"],3), r["roofline"]["kernel"])'
for o in sampled grouped; do
  $B --pair-order $o --variant 2 | python -c "$P" "v2 $o"
  for st in 1 2 4; do $B --pair-order $o --segment-steps $st | python -c "$P" "segment $st $o"; done
done
for d in 32 64 96 256 512; do
  $B --dim $d --pair-order grouped --variant 2 | python -c "$P" "dim $d v2 grouped"
  $B --dim $d --pair-order sampled --variant 2 | python -c "$P" "dim $d v2 sampled"
  for st in 1 2 4; do
    $B --dim $d --pair-order grouped --segment-steps $st | python -c "$P" "dim $d segment $st grouped"
    $B --dim $d --pair-order sampled --segment-steps $st | python -c "$P" "dim $d segment $st sampled"
  done
done
