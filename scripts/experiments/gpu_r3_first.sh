# Round 3, first GPU job: (A) learning quality at P = 4 / 8 / 16 on "hub100k" against the reference's goldens (0.902-0.904 at
# every P), per kernel / update mode; (B) what atomic delta updates cost; (C) the short form of the bench.
set -x
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
E=scripts/experiments/auc_shapes.py
{
for conf in "partitions=16 episode=2" "partitions=8 episode=5" "partitions=4 episode=9"; do
  for mode in "" "atomic=1" "atomic=3" "variant=2 atomic=3" "variant=2" "generation=5120" "generation=1024"; do
    timeout 300 python $E hub100k 200 auto 17,18 $conf $mode 2>&1 | grep -E "mean|Error|error"
  done
done
timeout 300 python $E hub100k 200 auto 17,18 atomic=3 2>&1 | grep -E "mean|Error"
timeout 300 python $E hub100k 200 auto 17,18 partitions=16 episode=2 device_sampling=1 atomic=1 2>&1 | grep -E "mean|Error"
} > gpurun_out/r3_partitions.txt 2>&1
{
Q="--no-cpu-baseline --no-end-to-end"
for t in "" "--tune 7=1" "--tune 7=3"; do
  python bench.py --steps 20 --warmup 5 $Q $t
  python bench.py --steps 400 --warmup 50 $Q $t
  python bench.py --steps 400 --warmup 50 $Q --partitions 16 $t
done
} > gpurun_out/r3_atomic_cost.jsonl 2> gpurun_out/r3_atomic_cost.err
timeout 300 python scripts/experiments/short_form.py > gpurun_out/r3_short_form.txt 2>&1
tail -30 gpurun_out/r3_partitions.txt
