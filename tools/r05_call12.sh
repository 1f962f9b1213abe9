cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export JAMIE_LIB=$PWD/jamie_amd/libjamie_hip_bnstamp.so
python tools/stamp_bn.py > gpurun_out/r05/stamps_bn_fwd_panel.log 2>&1
python -c "
import sys; sys.path.insert(0, '.')
from jamie_amd import engine
engine.tune(bn_panel=False)
exec(open('tools/stamp_bn.py').read())
" > gpurun_out/r05/stamps_bn_fwd_rowmajor.log 2>&1
tail -22 gpurun_out/r05/stamps_bn_fwd_panel.log; echo ------; tail -14 gpurun_out/r05/stamps_bn_fwd_rowmajor.log
