set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r05/base_bench.json 2> gpurun_out/r05/base_bench.err
tail -c 600 gpurun_out/r05/base_bench.json; echo
export JAMIE_HIP_LIB=$PWD/tools/libjamie_stamp.so
CASES=fwd_d2d CFG=31 python tools/stamp_gemm_bf16.py > gpurun_out/r05/stamps_fwd.log 2>&1
CASES=fwd_2dd CFG=32 python tools/stamp_gemm_bf16.py >> gpurun_out/r05/stamps_fwd.log 2>&1
CASES=bwd_now python tools/stamp_gemm_bf16.py >> gpurun_out/r05/stamps_fwd.log 2>&1
unset JAMIE_HIP_LIB
cat gpurun_out/r05/stamps_fwd.log
bash tools/ab_flags.sh -r 2 "--dtype f32" "--dtype f32 --pipeline" > gpurun_out/r05/ab_f32_pipeline.log 2>&1
cat gpurun_out/r05/ab_f32_pipeline.log
