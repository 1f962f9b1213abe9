cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -q -m gpu > gpurun_out/r05/t_all_gpu.log 2>&1
echo "pytest rc $?"
tail -5 gpurun_out/r05/t_all_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
