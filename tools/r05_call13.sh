set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "panel or bn_act or mse" > gpurun_out/r05/t_panel8.log 2>&1 || (tail -40 gpurun_out/r05/t_panel8.log; exit 1)
tail -2 gpurun_out/r05/t_panel8.log
python -m pytest tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r05/t_configs3.log 2>&1 || (tail -60 gpurun_out/r05/t_configs3.log; exit 1)
tail -2 gpurun_out/r05/t_configs3.log
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_p16.so" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_p8cq4.so" > gpurun_out/r05/ab_panel8.log 2>&1
cat gpurun_out/r05/ab_panel8.log
