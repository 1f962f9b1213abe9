cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_hip_configs.py tests/test_hip_step.py tests/test_hip_kernels.py -q -m gpu -x > gpurun_out/r05/t_latpin.log 2>&1
rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r05/t_latpin.log
[ $rc -eq 0 ] && bash tools/ab.sh -r 4 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_nopin.so" > gpurun_out/r05/ab_latent_descriptor_burst.log 2>&1
cat gpurun_out/r05/ab_latent_descriptor_burst.log
