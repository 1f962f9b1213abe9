#!/bin/bash
# GPU box: rocprofv3 --kernel-trace of a short bf16 bench run; average duration per (kernel, grid size): tells the launches of one
# kernel apart (BatchNorm over 2d-wide vs d-wide layers, the forward GEMMs, ...).   tools/prof_by_grid.sh TAG [ENV=..]...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-f32-record > $OUT/bench.json 2> $OUT/err.log
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/trace/*/*_kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0][-70:]
    acc[(name, r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for (name, grid, wg), v in rows[:28]:
    v = v[len(v) // 4:]
    print(f'{name:70s} grid {grid:>8s} wg {wg:>5s}  n {len(v):4d}  avg {sum(v)/len(v):7.1f} us  min {min(v):7.1f}')
PY
rm -rf $OUT/trace
