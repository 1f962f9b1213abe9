set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_r4gemm.so" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_bnsm.so" > gpurun_out/r05/ab_r4gemm_bnsm.log 2>&1
cat gpurun_out/r05/ab_r4gemm_bnsm.log
