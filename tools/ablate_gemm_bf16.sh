#!/bin/bash
# Diagnostic (GPU box): rebuild with one phase of the bf16 GEMM removed; timings only (results are wrong).
set -e
cd "$(dirname "$0")/.."
for abl in 1 2 3 4 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DJAMIE_GEMMB_ABL=$abl -o jamie_amd/libjamie_hip.so jamie_amd/csrc/*.hip
  echo "=== ABL $abl"
  NBUF=${NBUF:-1} CFGS=${1:-1} python tools/bench_gemm_bf16.py 2>&1 | grep -E "splitk 1"
done
