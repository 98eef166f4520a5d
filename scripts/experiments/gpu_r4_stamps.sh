#!/bin/bash
# Round 4: the stamps of isolated train_hot_kernel launches (timestamp library) on the headline shape and at the shard size of
# an 8-GPU run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/stamps
export GVK_ALLOW_TEST_LIBRARY=1 GVK_LIBRARY=graphvite_amd/csrc/build/ts/libgvk_ts.so
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-end-to-end --no-module --no-access-pattern"
GVK_STAMP_FILE=/tmp/st_p1 timeout 200 $B $EXTRA > gpurun_out/stamps/p1.json 2> gpurun_out/stamps/p1.err
python scripts/experiments/stamps.py /tmp/st_p1.40 /tmp/st_p1.55 /tmp/st_p1.63 | tee gpurun_out/stamps/p1.txt
GVK_STAMP_FILE=/tmp/st_p8 timeout 200 $B --partitions 8 $EXTRA > gpurun_out/stamps/p8.json 2> gpurun_out/stamps/p8.err
python scripts/experiments/stamps.py /tmp/st_p8.40 /tmp/st_p8.63 | tee gpurun_out/stamps/p8.txt
cp /tmp/st_p1.55 /tmp/st_p8.63 gpurun_out/stamps/ 2>/dev/null
tail -n 3 gpurun_out/stamps/p1.err
