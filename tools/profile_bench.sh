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
timeout -k 10 500 python $ROOT/bench.py --dtype $DT --steps ${STEPS:-200} --warmup 20 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done" > $OUT/progress.txt
# (the profiled passes run the headline leg only: the fp32 record and the other configurations would mix their launches of the
#  same kernels into the per-kernel averages)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --dtype $DT --steps 50 --warmup 10 --no-cpu-baseline --no-f32-record --no-other-configs > $OUT/stats.json 2> /dev/null
echo "stats done" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-f32-record --no-other-configs > /dev/null 2>&1
echo "fetch done" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-f32-record --no-other-configs > /dev/null 2>&1
# the kernel trace of the stats pass (start / end of every dispatch: gaps between launches) is kept as a compact table of the
# LAST 12 steps; the full traces are large
python $ROOT/tools/trace_gaps.py $OUT > $OUT/launch_gaps.txt 2>&1 || true
find $OUT -name '*_kernel_trace.csv' -path '*stats*' -delete
ls -R $OUT | head -30
cat $OUT/bench.json
