cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/debug_x3_step.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_x3_step_debug.log
