#!/bin/bash
# GPU box: true kernel durations (rocprofv3 kernel trace) of tools/bench_bn.py, median per case; optional ablation builds
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
for abl in ${1:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DJAMIE_BN_ABL=$abl -o jamie_amd/libjamie_hip.so jamie_amd/csrc/*.hip
  OUT=$ROOT/gpurun_out/trace_bn_$abl; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $ROOT/tools/bench_bn.py > $OUT/log.txt 2>&1)
  echo "=== ABL $abl"
  python - <<PY
import csv, glob, statistics
f = glob.glob('$OUT/t/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'bn_act' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
names = open('$OUT/log.txt').read().splitlines()
names = [l.split(':')[0] for l in names if l.startswith('N=')]
n = 53
for i, nm in enumerate(names):
    seg = d[i * n + 3:(i + 1) * n]
    if seg: print(f'{nm}: kernel median {statistics.median(seg):6.1f} us')
PY
done
