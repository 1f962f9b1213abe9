cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 2 "-" "JAMIE_TUNE=f32_rows_cfg=20+f32_dw_cfg=20" "JAMIE_TUNE=f32_rows=20:3,2;20:3,2+f32_dw_cfg=20" "JAMIE_TUNE=f32_rows=20:4,4;20:4,4+f32_dw_cfg=20" "JAMIE_TUNE=f32_rows=20:2,1;20:2,1+f32_dw_cfg=20" "JAMIE_TUNE=f32_rows=20:4,4;20:6,3+f32_dw_cfg=20" -- --dtype f32 > gpurun_out/r05/ab_f32_x3.log 2>&1
cat gpurun_out/r05/ab_f32_x3.log
