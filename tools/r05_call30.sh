cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_f32stamp.so CFGROWS=20 CFGDW=20 CASES=fwd_d2d,dx_K2d,dw_4 timeout -k 10 300 python tools/stamp_gemm_f32.py > gpurun_out/r05/stamps_f32_x3.log 2>&1
grep -v amdgpu.ids gpurun_out/r05/stamps_f32_x3.log | head -60
