set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "mse or latent" > gpurun_out/r05/t_k2.log 2>&1 || (tail -40 gpurun_out/r05/t_k2.log; exit 1)
tail -2 gpurun_out/r05/t_k2.log
python -m pytest tests/test_hip_configs.py -x -q -m gpu > gpurun_out/r05/t_configs2.log 2>&1 || (tail -60 gpurun_out/r05/t_configs2.log; exit 1)
tail -2 gpurun_out/r05/t_configs2.log
bash tools/ab.sh -r 3 "-" "JAMIE_TUNE=fused_heads=False" "JAMIE_TUNE=mse_colpart=False" > gpurun_out/r05/ab_fused_heads.log 2>&1
cat gpurun_out/r05/ab_fused_heads.log
