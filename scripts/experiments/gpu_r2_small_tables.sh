set -x
E=scripts/experiments/auc_shapes.py
for rep in 1 2; do
python $E blog 2000 grouped 17,18,19,20,21 steps=4 2>&1 | grep -E "mean|Error"
python $E blog 2000 grouped 17,18,19,20,21 variant=4 2>&1 | grep -E "mean|Error"
python $E blog 2000 grouped 17,18,19,20,21 variant=4 run_cap=32 2>&1 | grep -E "mean|Error"
done
python $E blog 2000 grouped 17,18,19,20,21 variant=4 run_cap=12 2>&1 | grep -E "mean|Error"
