#!/bin/bash
# GPU box: A/B/... of bench configurations on the SAME box, interleaved (boxes of the pool differ by 3-6 %):
#   tools/ab.sh [-r ROUNDS] "ENV_A=.." "ENV_B=.." ["ENV_C=.." ...] [-- bench args]
# Every variant is a string of environment assignments ("X=1 Y=2"; "-" for none).  Legacy form: tools/ab.sh A B ROUNDS [args].
set -u
R=3
if [ "${1:-}" = "-r" ]; then R="$2"; shift 2; fi
V=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  if [[ "$1" =~ ^[0-9]+$ ]] && [ ${#V[@]} -ge 2 ]; then R="$1"; shift; break; fi     # legacy: the third argument is the round count
  V+=("$1"); shift
done
if [ "${1:-}" = "--" ]; then shift; fi
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in $(seq 1 $R); do
  n=0
  for E in "${V[@]}"; do
    n=$((n + 1))
    EE="$E"; if [ "$E" = "-" ]; then EE=""; fi
    out=$(env $EE python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-f32-record --no-other-configs "$@" 2>/dev/null | tail -1)
    echo "v$n [$E] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_event_timing_ms',{}); print(round(d['value']), round(1e3*d['ms_per_step'],1), 'us/step  adam', round(1e3*d['roofline']['avg_launch_ms'],1), ' enc_gemm', round(1e3*k.get('enc_gemm',{}).get('median',0),1))")"
  done
done
