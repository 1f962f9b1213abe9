#!/bin/bash
# GPU box: diagnostic build of the latent kernels with in-kernel timestamps (JAMIE_LAT_STAMP), microbenchmark, product rebuild
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
JAMIE_HIPCC_FLAGS="-DJAMIE_LAT_STAMP" python -c "import jamie_amd.build as b; b.build_library(force=False)" 
python tools/stamp_latent.py
# back to the product build (the recorded flag string differs: everything is recompiled)
python -c "import jamie_amd.build as b; b.build_library()"
