#!/bin/bash
# Build the stamped diagnostic library next to (not over) the product library (hipcc cross-compiles without a GPU);
# on the GPU box:  JAMIE_HIP_LIB=$PWD/tools/libjamie_stamp.so python tools/stamp_gemm_bf16.py
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -instcombine-max-copied-from-constant-users=100000 -DJAMIE_GEMMB_STAMP -DJAMIE_EXPERIMENTS $EXTRA -o tools/libjamie_stamp.so jamie_amd/csrc/*.hip
