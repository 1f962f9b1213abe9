cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "gemm_f32" > gpurun_out/r05/t_x3.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -25 gpurun_out/r05/t_x3.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_gemm.py 17,20 > gpurun_out/r05/bench_gemm_x3.log 2>&1
cat gpurun_out/r05/bench_gemm_x3.log
