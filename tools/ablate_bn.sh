#!/bin/bash
# Diagnostic (GPU box): float4 BN forward kernel with loads / stores removed (timing only)
cd "$(dirname "$0")/.."
for abl in 3 1 2 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DJAMIE_BN_ABL=$abl -o jamie_amd/libjamie_hip.so jamie_amd/csrc/*.hip
  echo "=== ABL $abl"
  python tools/bench_bn.py 2>&1 | grep "fwd"
done
