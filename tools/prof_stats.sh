#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of a short bf16 bench run; prints the per-kernel table (name, calls, average us).
#   tools/prof_stats.sh TAG [ENV=..]...      (BENCH_ARGS="--dtype f32": more bench.py arguments)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-f32-record ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/err.log
find $OUT -name '*_kernel_trace.csv' -delete
python - <<PY
import csv, glob
f = glob.glob('$OUT/stats/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
steps = 70.0
tot = 0
for r in rows:
    per = float(r['TotalDurationNs']) / 1e3 / steps
    tot += per
    if per > 0.5:
        print(f"{r['Name'][:100]:100s} {float(r['Calls'])/steps:5.1f}/step {float(r['AverageNs'])/1e3:8.1f} us  {per:7.1f} us/step")
print('sum of kernel time per step: %.1f us' % tot)
PY
