#!/bin/bash
# GPU box: rocprofv3 kernel stats of a short bench run only (no PMC passes, no CPU baseline / fp32 leg): gpurun_out/$1/stats.csv
set -u
TAG=${1:-q}
DT=${2:-bf16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --dtype $DT --steps 60 --warmup 10 --no-cpu-baseline --no-f32-record ${EXTRA:-} > $OUT/stats.json 2> /dev/null
find $OUT -name '*_kernel_trace.csv' -delete
cp $OUT/stats/*/*_kernel_stats.csv $OUT/stats.csv
python - "$OUT/stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = max(int(r['Calls']) for r in rows if 'clip_adam' in r['Name'])
tot = 0
for r in rows[:34]:
    per_step = float(r['TotalDurationNs']) / calls / 1e3
    tot += per_step
    print(f"{r['Name'][:70]:70s} {int(r['Calls'])/calls:5.1f}/step {float(r['AverageNs'])/1e3:8.1f} us  {per_step:7.1f} us/step")
print('sum us/step', round(tot, 1))
PY
