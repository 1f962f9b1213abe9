set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "mse" > gpurun_out/r05/t_mse.log 2>&1 || (tail -40 gpurun_out/r05/t_mse.log; exit 1)
tail -2 gpurun_out/r05/t_mse.log
python -m pytest tests/test_hip_configs.py tests/test_hip_step.py -x -q -m gpu > gpurun_out/r05/t_step.log 2>&1 || (tail -40 gpurun_out/r05/t_step.log; exit 1)
tail -2 gpurun_out/r05/t_step.log
bash tools/ab.sh -r 3 "-" "JAMIE_TUNE=mse_colpart=False" > gpurun_out/r05/ab_mse_colpart.log 2>&1
cat gpurun_out/r05/ab_mse_colpart.log
bash tools/ab.sh -r 2 "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_exp.so JAMIE_BN_CQ_FWD=4" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_exp.so JAMIE_BN_CQ_FWD=8" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_exp.so JAMIE_BN_CQ_FWD=2" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_exp.so JAMIE_BN_CQ_BWD=8" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_exp.so JAMIE_BN_CQ_BWD=2" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_exp.so" > gpurun_out/r05/ab_bn_cq_panel.log 2>&1
cat gpurun_out/r05/ab_bn_cq_panel.log
