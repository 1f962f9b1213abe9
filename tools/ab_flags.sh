#!/bin/bash
# GPU box: interleaved A/B of bench.py FLAG sets on one box:  tools/ab_flags.sh [-r ROUNDS] "" "--pipeline" "--pipeline --opt-priority -1"
set -u
R=2
if [ "${1:-}" = "-r" ]; then R="$2"; shift 2; fi
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in $(seq 1 $R); do
  for F in "$@"; do
    out=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-f32-record --no-other-configs $F 2>/dev/null | tail -1)
    echo "[$F] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],1), 'us/step')")"
  done
done
