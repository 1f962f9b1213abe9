cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for c in 17 20 17 20; do echo "EVAL_CFG=$c"; EVAL_CFG=$c timeout -k 10 200 python tools/bench_infer.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05/bench_infer_x3.log 2>&1
cat gpurun_out/r05/bench_infer_x3.log
timeout -k 10 400 python -m pytest tests/test_hip_step.py -q -m gpu -x -k "bf16_pipe_stays" 2>&1 | tail -3
