#!/bin/bash
# GPU box: fp32 step time vs the tile configuration of the dW (TN) launches and of the unsplit dX (NN) launches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
run() {
  out=$(env $1 python bench.py --dtype f32 --steps 100 --warmup 10 --no-cpu-baseline --no-f32-record 2>/dev/null | tail -1)
  echo "[$1] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],1), 'us/step')")"
}
for r in 1 2; do
  for c in 1 10 11 12 4 0 3; do run "JAMIE_TUNE=f32_dw_cfg=$c"; done
done
for c in 1 10 11 12 4; do run "JAMIE_TUNE=f32_dx_cfg=$c"; done
