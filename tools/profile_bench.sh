#!/bin/bash
# GPU box: bench + rocprofv3 kernel stats + HBM traffic counters (separate --pmc passes, as
# /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes).  Outputs land in gpurun_out/$1/.
set -u
TAG=${1:-prof}
DT=${2:-bf16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --dtype $DT --steps ${STEPS:-200} --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --dtype $DT --steps 50 --warmup 10 --no-cpu-baseline > $OUT/stats.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
# keep only the small summaries (the traces are large)
find $OUT -name '*_kernel_trace.csv' -path '*stats*' -delete
ls -R $OUT | head -30
cat $OUT/bench.json
