cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "mse" 2>&1 | tail -2
python tools/bench_mse.py | tail -4
for V in True False; do
OUT=$PWD/gpurun_out/r05prof_mse_$V
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && JAMIE_TUNE=mse_colpart=$V rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-record --no-other-configs > $OUT/stats.json 2> /dev/null )
python tools/trace_gaps.py $OUT | grep -E "steps of|mse_cast|bn_act_bwd4" | head -3
find $OUT -name '*_kernel_trace.csv' -delete
done
