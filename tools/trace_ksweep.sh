#!/bin/bash
# GPU box: true kernel durations (rocprofv3 kernel trace) of the K sweep -> median per K, per config
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/trace_ksweep; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp CFGS=${1:-7,20}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $ROOT/tools/bench_gemm_bf16_ksweep.py > $OUT/log.txt 2>&1)
python - <<PY
import csv, glob, statistics
f = glob.glob('$OUT/t/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'gemm_bf16' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
gap = [(int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp'])) / 1e3 for i in range(len(rows) - 1)]
n = 51   # 3 warm-up + 48 timed launches per (cfg, K)
cfgs = '$CFGS'.split(',')
Ks = (64, 128, 256, 512, 1024, 2048, 4096)
for ci, c in enumerate(cfgs):
    for ki, K in enumerate(Ks):
        s = (ci * len(Ks) + ki) * n
        seg, gs = d[s + 3:s + n], gap[s + 3:s + n - 1]
        print(f'cfg {c} K {K:5d}: kernel median {statistics.median(seg):7.1f} us   gap median {statistics.median(gs):6.1f} us')
PY
