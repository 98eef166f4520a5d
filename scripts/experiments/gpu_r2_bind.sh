set -x
python -m pytest tests/test_bind_gpu.py tests/test_bind_cpu.py -x -q -s 2>&1 | tail -25
python -m pytest tests/test_kernel_gpu.py -x -q 2>&1 | tail -3
E=scripts/experiments/auc_shapes.py
python $E blog 2000 sampled,grouped 17,18,19,20,21 variant=2 2>&1 | grep mean
python $E blog 2000 sampled,grouped 17,18,19,20,21 steps=1 2>&1 | grep mean
python $E blog 2000 sampled,grouped 17,18,19,20,21 steps=2 2>&1 | grep mean
python $E blog 2000 sampled,grouped 17,18,19,20,21 steps=4 2>&1 | grep mean
python $E blog 2000 grouped 17,18,19,20,21 variant=4 2>&1 | grep mean
python $E hub100k 200 sampled,grouped 17,18,19,20,21 steps=4 2>&1 | grep mean
python $E hub100k 200 sampled,grouped 17,18,19,20,21 steps=1 2>&1 | grep mean
