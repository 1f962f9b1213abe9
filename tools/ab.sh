#!/bin/bash
# GPU box: A/B of two bench configurations on the SAME box, interleaved (boxes of the pool differ by 3-6 %):
#   tools/ab.sh "ENV_A=.. " "ENV_B=.." [rounds] [bench args]
set -u
A="$1"; B="$2"; R=${3:-3}; shift 3 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-f32-record "$@" 2>/dev/null | tail -1)
    echo "$v [$E] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],1), 'us/step  adam', round(1e3*d['roofline']['avg_launch_ms'],1))")"
  done
done
