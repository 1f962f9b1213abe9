cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 300 python tools/debug_x3_determinism.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/x3_determinism.log
