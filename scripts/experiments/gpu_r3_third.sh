# Round 3, third GPU job: equal parts of a batch, each regrouped on its own and trained by its own launch
# (gvk_train_launches), against the reference's goldens at P = 4 / 8 / 16.
set -x
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernel_gpu.py -q -x -k "several_launches or runs or group" 2>&1 | tail -5 > gpurun_out/r3_kernel_tests.txt
E=scripts/experiments/auc_shapes.py
{
for conf in "partitions=16 episode=2" "partitions=8 episode=5" "partitions=4 episode=9"; do
  for mode in "split=2" "split=1" "split=4"; do
    timeout 300 python $E hub100k 200 auto 17,18 $conf $mode 2>&1 | grep -E "mean|Error|error"
  done
done
timeout 300 python $E hub100k 200 auto 17,18 partitions=16 episode=2 device_sampling=1 2>&1 | grep -E "mean|Error"
timeout 300 python $E hub100k 200 auto 17,18 2>&1 | grep -E "mean|Error"
timeout 300 python $E blog 2000 auto 17,18 2>&1 | grep -E "mean|Error"
} > gpurun_out/r3_partitions_parts.txt 2>&1
cat gpurun_out/r3_kernel_tests.txt
grep mean gpurun_out/r3_partitions_parts.txt
