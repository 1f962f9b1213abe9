set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 2 "-" "JAMIE_TUNE=bwd_k_per_slab=1000" "JAMIE_TUNE=bwd_k_per_slab=1400" "JAMIE_TUNE=bf16_rows=31:4,2;32:3,2" "JAMIE_TUNE=bf16_rows=31:3,2;32:4,2" "JAMIE_TUNE=sk_skinny=4" > gpurun_out/r05/ab_slab_plans_panel.log 2>&1
cat gpurun_out/r05/ab_slab_plans_panel.log
