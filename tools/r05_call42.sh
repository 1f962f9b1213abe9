cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 2 "-" "JAMIE_TUNE=f32_rows=21:2,1;21:4,2" "JAMIE_TUNE=f32_rows=21:4,2;21:8,4" "JAMIE_TUNE=f32_rows=21:3,1;21:6,3" "JAMIE_TUNE=f32_rows=20:2,1;20:4,2+f32_dw_cfg=20" -- --dtype f32 --config c5dims --steps 60 --warmup 10 > gpurun_out/r05/ab_f32_x3_256_c5dims.log 2>&1
cat gpurun_out/r05/ab_f32_x3_256_c5dims.log
for c in 20 21; do echo "EVAL_CFG=$c"; EVAL_CFG=$c timeout -k 10 200 python tools/bench_infer.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05/bench_infer_x3_256.log 2>&1
cat gpurun_out/r05/bench_infer_x3_256.log
