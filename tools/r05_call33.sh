cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_hip_step.py tests/test_hip_configs.py tests/test_hip_blocks.py -q -m gpu -x > gpurun_out/r05/t_step_x3.log 2>&1
echo rc $?
grep -v amdgpu gpurun_out/r05/t_step_x3.log | tail -8
