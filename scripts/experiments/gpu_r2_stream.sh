set -x
python -m pytest tests/test_kernel_gpu.py -x -q 2>&1 | tail -3
B='python bench.py --no-cpu-baseline --no-end-to-end'
P='import json,sys; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r["value"]), round(r["roofline"]["kernel_ms"]*1e3,2), round(r["roofline"]["frac"],3), r["roofline"]["kernel"])'
for o in grouped sampled; do
  $B --pair-order $o --variant 2 | python -c "$P" "v2 $o"
  $B --pair-order $o --segment-steps 1 | python -c "$P" "seg1 $o"
  $B --pair-order $o --segment-steps 1 --tune 9=1 | python -c "$P" "seg1 nt-stores $o"
  for st in 2 4 8; do
    $B --pair-order $o --segment-steps $st --tune 8=1 | python -c "$P" "stream $st $o"
  done
  $B --pair-order $o --segment-steps 4 --tune 8=1 --tune 9=1 | python -c "$P" "stream 4 nt $o"
done
for d in 32 64 96; do
  $B --dim $d --pair-order sampled --variant 2 | python -c "$P" "dim $d v2 sampled"
  for st in 2 4 8; do $B --dim $d --pair-order sampled --segment-steps $st --tune 8=1 | python -c "$P" "dim $d stream $st sampled"; done
  $B --dim $d --pair-order grouped --segment-steps 4 --tune 8=1 | python -c "$P" "dim $d stream 4 grouped"
done
