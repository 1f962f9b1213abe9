set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "panel or bn_act" > gpurun_out/r05/t_panel.log 2>&1 || (tail -40 gpurun_out/r05/t_panel.log; exit 1)
tail -2 gpurun_out/r05/t_panel.log
python -m pytest tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r05/t_configs.log 2>&1 || (tail -40 gpurun_out/r05/t_configs.log; exit 1)
tail -2 gpurun_out/r05/t_configs.log
bash tools/ab.sh -r 3 "-" "JAMIE_TUNE=bn_panel=False" > gpurun_out/r05/ab_bn_panel.log 2>&1
cat gpurun_out/r05/ab_bn_panel.log
