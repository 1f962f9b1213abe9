for v in "" "32:1,1" "29:1,1" "29:2,1;29:3,2" "32:2,1;32:3,2" "31:2,1;32:4,2" "31:3,2;32:6,3" "31:4,2;32:8,4"; do
  out=$(python bench.py --tune "bf16_rows=$v" --steps 200 --warmup 20 --no-cpu-baseline --no-f32-record 2>/dev/null | tail -1)
  echo "rows[$v] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],1), 'us/step  adam', round(1e3*d['roofline']['avg_launch_ms'],1), ' fwd gemm', round(1e3*d['roofline']['gemm_avg_launch_ms'],1))")"
done
