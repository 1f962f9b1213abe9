set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
JAMIE_HIP_LIB=$PWD/tools/libjamie_stamp_base.so python tools/stamp_step_bf16.py > gpurun_out/r05/stamps_step_base.log 2>&1
JAMIE_HIP_LIB=$PWD/tools/libjamie_stamp_pb.so python tools/stamp_step_bf16.py > gpurun_out/r05/stamps_step_pb.log 2>&1
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_pb.so" > gpurun_out/r05/ab_pro_barrier.log 2>&1
cat gpurun_out/r05/ab_pro_barrier.log
