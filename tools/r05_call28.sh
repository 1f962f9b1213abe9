cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/profile_bench.sh r05prof_f32_v2 f32 > gpurun_out/r05/prof_f32_v2.log 2>&1
tail -3 gpurun_out/r05/prof_f32_v2.log | cut -c1-1500
