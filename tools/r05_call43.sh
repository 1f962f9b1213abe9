cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_step.py -q -m gpu -x 2>&1 | grep -v amdgpu | tail -4
