cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "gemm_f32" > gpurun_out/r05/t_x3_256.log 2>&1
rc=$?; echo "pytest rc $rc"; grep -v amdgpu gpurun_out/r05/t_x3_256.log | tail -12
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_gemm.py 20,21 2>&1 | grep -v amdgpu > gpurun_out/r05/bench_gemm_x3_256.log
cat gpurun_out/r05/bench_gemm_x3_256.log
