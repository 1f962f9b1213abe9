cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/profile_bench.sh r05prof_f32_v3 f32 > gpurun_out/r05/prof_f32_v3.log 2>&1
tail -1 gpurun_out/r05/prof_f32_v3.log | cut -c1-300
( time timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_style_v3.json 2> gpurun_out/r05/bench_driver_style_v3.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_driver_style_v3.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), d['ms_per_step'], 'steady', d.get('steady_state',{}).get('ms_per_step'))
f=d.get('f32',{}); print('f32', f.get('value'), f.get('ms_per_step'), f.get('error'), f.get('roofline',{}).get('frac'), f.get('roofline',{}).get('avg_launch_ms'))
for k,v in d.get('other_configs',{}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
PY
