cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 --step-trace --no-cpu-baseline --no-f32-record --no-other-configs > gpurun_out/r05/step_trace.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/step_trace.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["steady_state"]["ms_per_step"])
t = d["step_trace_us"]
print(t[:40]); print(t[40:100]); print(t[-20:])
PY
