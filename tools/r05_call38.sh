cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 3 "-" "JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_x3h16.so" -- --dtype f32 > gpurun_out/r05/ab_f32_x3_lds_halves.log 2>&1
cat gpurun_out/r05/ab_f32_x3_lds_halves.log
