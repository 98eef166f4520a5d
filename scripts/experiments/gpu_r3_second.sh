# Round 3, second GPU job: the batch split over launches (GVK_TUNE_SPLIT_HITS) against the reference's goldens at
# P = 4 / 8 / 16, the kernel tests, and the bench after the load reordering.
set -x
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernel_gpu.py -q -x 2>&1 | tail -8 > gpurun_out/r3_kernel_tests.txt
E=scripts/experiments/auc_shapes.py
{
for conf in "partitions=16 episode=2" "partitions=8 episode=5" "partitions=4 episode=9"; do
  for mode in "split=4" "split=2" "split=8" "split=1"; do
    timeout 300 python $E hub100k 200 auto 17,18 $conf $mode 2>&1 | grep -E "mean|Error|error"
  done
done
timeout 300 python $E hub100k 200 auto 17,18 partitions=16 episode=2 split=4 device_sampling=1 2>&1 | grep -E "mean|Error"
timeout 300 python $E blog 2000 auto 17,18 2>&1 | grep -E "mean|Error"
} > gpurun_out/r3_partitions_split.txt 2>&1
{
Q="--no-cpu-baseline --no-end-to-end"
python bench.py --steps 20 --warmup 5 $Q
python bench.py --steps 400 --warmup 50 $Q
python bench.py --steps 20 --warmup 5 $Q
python bench.py --steps 400 --warmup 50 $Q
python bench.py --steps 400 --warmup 50 $Q --partitions 16
python bench.py --steps 400 --warmup 50 $Q --partitions 8
for d in 64 96; do python bench.py --dim $d $Q --steps 400 --warmup 50; done
} > gpurun_out/r3_bench_reorder.jsonl 2> gpurun_out/r3_bench_reorder.err
cat gpurun_out/r3_kernel_tests.txt
grep mean gpurun_out/r3_partitions_split.txt
