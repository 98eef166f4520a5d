cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python scripts/experiments/configs_auc.py job=fs_line_p8 seeds=1024,5 variants="device=1;device=1,tune=12:4;device=1,tune=12:2;device=1,tune=12:1;device=1,parts=50" 2>&1 | grep -v amdgpu.ids | grep -E "^fs|Error|error" | tee $O/r5_fs_device_rounds.log
