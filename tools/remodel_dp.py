"""Re-price the saved timelines of bench.py --dry-run-world runs (profiles/r03_dp_model_*.json) with bench.dp_model as it stands
now (assumed bus bandwidth / latencies changed: no GPU needed, the HIP-event timeline is in the file).
Usage: python tools/remodel_dp.py profiles/r03_dp_model_*.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    m = d['dp_model']
    ev = [(e['kind'], e['bytes'], e['at_us'] * 1e-3) for e in m['events']]
    if 'traced_step_us' in m:
        ev.append(('step_end', 0, m['traced_step_us'] * 1e-3))
    bf = d['dtype'] == 'bf16'          # (the message dtype of these runs is the compute dtype)
    new = bench.dp_model(ev, m['step_us_dry_run'] * 1e-3, m['step_us_one_gpu'] * 1e-3, m['world'], 2.0 if bf else 0.5,
                         'fp32_messages' if bf else 'bf16_messages')
    d['dp_model'] = new
    open(path, 'w').write(json.dumps(d) + '\n')
    print(os.path.basename(path), m['optimizer'], {k: (round(v['predicted_step_us'], 1), round(v['predicted_speedup'], 2), {a: round(b) for a, b in v['stalls'].items()})
                                                   for k, v in new.items() if isinstance(v, dict) and 'predicted_step_us' in v})
