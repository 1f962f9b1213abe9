set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -x -q -m gpu > gpurun_out/r05/t_kernels.log 2>&1 || (tail -30 gpurun_out/r05/t_kernels.log; exit 1)
tail -2 gpurun_out/r05/t_kernels.log
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_soff.so" > gpurun_out/r05/ab_store_voff_bf16.log 2>&1
cat gpurun_out/r05/ab_store_voff_bf16.log
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_soff.so" -- --dtype f32 > gpurun_out/r05/ab_store_voff_f32.log 2>&1
cat gpurun_out/r05/ab_store_voff_f32.log
