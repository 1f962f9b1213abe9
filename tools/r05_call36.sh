cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/pmc_step_f32.sh > gpurun_out/r05/pmc_step_f32_x3.txt 2>&1
cat gpurun_out/r05/pmc_step_f32_x3.txt | cut -c1-260
