cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_style_v2.json 2> gpurun_out/r05/bench_driver_style_v2.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_driver_style_v2.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), d['ms_per_step'], 'steady', d.get('steady_state',{}).get('ms_per_step'))
f=d.get('f32',{}); print('f32', f.get('value'), f.get('ms_per_step'), f.get('error'), f.get('child_process_seconds'), f.get('roofline',{}).get('frac'))
for k,v in d.get('other_configs',{}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'), v.get('child_process_seconds'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
timeout -k 10 500 python -m pytest tests/test_hip_step.py -q -m gpu -x -k "bench" 2>&1 | tail -3
