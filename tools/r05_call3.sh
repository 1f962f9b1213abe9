set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/ab.sh -r 2 "-" "JAMIE_TUNE=bf16_rows=31:2,1;32:3,2" "JAMIE_TUNE=bf16_rows=31:2,2;32:3,2" "JAMIE_TUNE=adam_rotate=True" > gpurun_out/r05/ab_fwd_slabs.log 2>&1
cat gpurun_out/r05/ab_fwd_slabs.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_new_line.json 2> gpurun_out/r05/bench_new_line.err || (tail -20 gpurun_out/r05/bench_new_line.err; exit 1)
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/bench_new_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('steady_state'))
print(json.dumps(d['roofline']['encoder_gemm'], indent=1))
print(json.dumps(d['f32']['roofline']['encoder_gemm'], indent=1))
print(json.dumps(d.get('other_configs'), indent=1))
PY
