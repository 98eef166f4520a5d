set -x
E=scripts/experiments/auc_shapes.py
python $E hub100k 200 auto 17,18,19 2>&1 | grep -E "mean|Error"
python $E hub100k 200 auto 17,18,19 partitions=4 episode=9 2>&1 | grep -E "mean|Error"
python $E hub100k 200 auto 17,18,19 partitions=8 episode=5 2>&1 | grep -E "mean|Error"
python $E hub100k 200 auto 17,18,19 partitions=16 episode=2 2>&1 | grep -E "mean|Error"
python $E hub100k 200 auto 17,18,19 partitions=4 episode=9 device_sampling=1 2>&1 | grep -E "mean|Error"
python $E hub100k 200 auto 17,18,19 partitions=16 episode=2 device_sampling=1 2>&1 | grep -E "mean|Error"
python $E hub100k 200 auto 17,18,19 device_sampling=1 2>&1 | grep -E "mean|Error"
