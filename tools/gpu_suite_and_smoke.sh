cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -q -m gpu > gpurun_out/r05/t_all_gpu_final.log 2>&1
echo "pytest rc $?"
tail -4 gpurun_out/r05/t_all_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
