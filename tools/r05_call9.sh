set -e
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r05prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-record --no-other-configs > $OUT/stats.json 2> /dev/null
python $GRAFT_REPO_ROOT/tools/trace_gaps.py $OUT > $OUT/launch_gaps.txt 2>&1 || true
cat $OUT/launch_gaps.txt
find $OUT -name '*_kernel_trace.csv' -delete
