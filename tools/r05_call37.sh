cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "gemm_f32" 2>&1 | tail -2
timeout -k 10 300 python tools/bench_gemm.py 20 2>&1 | grep -v amdgpu > gpurun_out/r05/bench_gemm_x3_v2.log; grep "splitk 1\|TN" gpurun_out/r05/bench_gemm_x3_v2.log
bash tools/pmc_step_f32.sh > gpurun_out/r05/pmc_step_f32_x3_v2.txt 2>&1
grep "gemm_f32_kernel\|conflict\|matrix pipe" gpurun_out/r05/pmc_step_f32_x3_v2.txt | cut -c1-200
