#!/bin/bash
# GPU box: the training step with alternative launch shapes of clip + Adam (libraries built by
# jamie_amd.build.build_variant('adam_U_T_GRID', ['-DJAMIE_ADAM_U=..', '-DJAMIE_ADAM_T=..', '-DJAMIE_ADAM_GRID=..'], only=['optim.hip'])),
# interleaved on one box, two rounds
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for r in 1 2; do
  for lib in jamie_amd/libjamie_hip.so jamie_amd/libjamie_hip_adam_*.so; do
    out=$(JAMIE_LIB=$ROOT/$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-f32-record 2>/dev/null | tail -1)
    echo "$lib $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],1), 'us/step  adam', round(1e3*d['roofline']['avg_launch_ms'],1))")"
  done
done
