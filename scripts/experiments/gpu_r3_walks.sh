# Round 3: DeepWalk / node2vec on "blog" against the reference's loop (0.8769 / 0.8762 / 0.8765) by pair order and kernel.
set -x
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
E=scripts/experiments/auc_shapes.py
{
for mode in "" "variant=2" "variant=4" "split=0" "variant=4 run_cap=5"; do
  for order in sampled grouped; do
    timeout 300 python $E blog 2000 $order 17,18 model=DeepWalk aug=5 episode=500 $mode 2>&1 | grep -E "mean|Error|error"
  done
done
timeout 300 python $E blog 2000 sampled 17,18 model=node2vec aug=5 p=0.25 q=0.25 episode=500 2>&1 | grep -E "mean|Error"
timeout 300 python $E blog 2000 sampled,grouped 17,18 2>&1 | grep -E "mean|Error"
} > gpurun_out/r3_walk_orders.txt 2>&1
timeout 300 python -m pytest tests/test_solver_gpu.py -q -s -k "walk_mode or parity_with_oracle" 2>&1 | grep -E "AUC|passed|failed" >> gpurun_out/r3_walk_orders.txt
grep -E "mean|AUC|passed" gpurun_out/r3_walk_orders.txt
