cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 2 "-" "JAMIE_TUNE=f32_rows=21:3,2;21:3,2" "JAMIE_TUNE=f32_rows=21:3,2;21:6,3" "JAMIE_TUNE=f32_rows=21:4,2;21:4,2" "JAMIE_TUNE=f32_rows=21:3,2;21:3,2+f32_dw_cfg=21" -- --dtype f32 > gpurun_out/r05/ab_f32_x3_256.log 2>&1
cat gpurun_out/r05/ab_f32_x3_256.log
