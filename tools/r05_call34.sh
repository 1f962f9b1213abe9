cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hip_step.py -q -m gpu -x -s -k "bf16_pipe_stays" 2>&1 | grep -v amdgpu | grep -A3 "per-matrix\|passed\|failed\|Error" | head -20
