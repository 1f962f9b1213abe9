#!/bin/bash
# Diagnostic (GPU box): rebuild the library with one GEMM phase removed and time the tile sweep.
# Outputs are WRONG in ablated builds; only the timings matter.  Restores the normal build at the end.
set -e
cd "$(dirname "$0")/.."
for abl in 1 2 4 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DJAMIE_GEMM_ABL=$abl -o jamie_amd/libjamie_hip.so jamie_amd/csrc/*.hip
  echo "=== ABL $abl"
  python tools/bench_gemm.py ${1:-1} 2>&1 | grep -v amdgpu.ids
done
