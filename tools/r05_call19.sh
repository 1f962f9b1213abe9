cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r05_bf16 bf16 > gpurun_out/r05_bf16_prof.log 2>&1
bash tools/profile_bench.sh r05_f32 f32 > gpurun_out/r05_f32_prof.log 2>&1
cat gpurun_out/r05_bf16/launch_gaps.txt | head -30
ls gpurun_out/r05_bf16 gpurun_out/r05_f32
