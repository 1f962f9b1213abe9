cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_style.json 2> gpurun_out/r05/bench_driver_style.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_driver_style.json").read().strip().splitlines()[-1])
print('headline', d["value"], d["ms_per_step"], 'steady', d["steady_state"]["ms_per_step"])
print('roofline', {k: v for k, v in d['roofline'].items() if k in ('frac', 'avg_launch_ms', 'whole_step_frac')})
e = d['roofline']['encoder_gemm']; print('enc0', e['avg_launch_ms'], e['frac_mfma'], e['frac_ingest'])
print('f32', d['f32']['value'], d['f32']['ms_per_step'], d['f32']['roofline']['frac'])
print('others', {k: (v.get('value'), v.get('ms_per_step')) for k, v in d['other_configs'].items() if isinstance(v, dict)})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['gpu_over_cpu'])
print(d['kernel_event_timing_ms'])
PY
