set -x
B='python bench.py --no-cpu-baseline --no-end-to-end --steps 1000 --warmup 100'
P='import json,sys; r=json.loads(sys.stdin.readline()); print(sys.argv[1], round(r["value"]), round(r["roofline"]["kernel_ms"]*1e3,2), round(r["roofline"]["frac"],3), r["roofline"]["kernel"])'
for rep in 1 2; do
$B --dim 32 | python -c "$P" "dim 32 default"
$B --dim 32 --lanes 4 | python -c "$P" "dim 32 lanes 4"
$B --dim 32 --lanes 16 | python -c "$P" "dim 32 lanes 16"
$B --dim 64 --pair-order sampled | python -c "$P" "dim 64 default"
$B --dim 64 --pair-order sampled --lanes 8 | python -c "$P" "dim 64 lanes 8"
$B --dim 64 --pair-order sampled --lanes 4 | python -c "$P" "dim 64 lanes 4"
$B --dim 96 | python -c "$P" "dim 96 default"
$B --dim 96 --lanes 16 | python -c "$P" "dim 96 lanes 16"
done
$B --dim 32 --batch 65536 | python -c "$P" "dim 32 batch 65536"
$B --dim 32 --batch 131072 | python -c "$P" "dim 32 batch 131072"
$B --dim 96 --batch 65536 | python -c "$P" "dim 96 batch 65536"
$B --dim 96 --batch 196608 | python -c "$P" "dim 96 batch 196608"
