set -x
B='python bench.py --no-cpu-baseline --no-end-to-end --steps 1000 --warmup 100'
P='import json,sys; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r["value"]), round(r["ms_per_step"]*1e3,2), round(r["roofline"]["kernel_ms"]*1e3,2), round(r["roofline"]["frac"],3), r["roofline"]["kernel"], r["config"]["block_batches"])'
for rep in 1 2; do
for parts in 16 8 4; do
  for o in sampled grouped; do
    $B --partitions $parts --pair-order $o | python -c "$P" "rep$rep P=$parts $o"
  done
done
done
$B --partitions 16 --pair-order grouped --segment-steps 4 | python -c "$P" "P=16 grouped seg4"
$B --partitions 16 --pair-order grouped --variant 2 | python -c "$P" "P=16 grouped v2"
