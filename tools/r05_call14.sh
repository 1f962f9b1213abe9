set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gemm_bf16 or panel or bn_act or mse" > gpurun_out/r05/t_k3.log 2>&1 || (tail -40 gpurun_out/r05/t_k3.log; exit 1)
tail -2 gpurun_out/r05/t_k3.log
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_nopre.so" > gpurun_out/r05/ab_kernarg_preload.log 2>&1
cat gpurun_out/r05/ab_kernarg_preload.log
