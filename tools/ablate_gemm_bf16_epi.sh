#!/bin/bash
# Diagnostic (GPU box): DMA bf16 GEMM with one phase removed (5: no output stores, 6: no fragment reads / MFMAs,
# 7: no LDS-DMA after the prologue), K sweep (fixed cost vs per-k-step cost).  Timings only.
cd "$(dirname "$0")/.."
for abl in ${2:-5 6 7 0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DJAMIE_GEMMB_ABL=$abl -o jamie_amd/libjamie_hip.so jamie_amd/csrc/*.hip
  echo "=== ABL $abl"
  CFGS=${1:-7,14,20} python tools/bench_gemm_bf16_ksweep.py 2>&1 | grep -E "K    64|K  2048|K  4096"
done
