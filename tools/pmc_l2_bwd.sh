#!/bin/bash
# GPU box: L2 hit rate (TCC_HIT_sum / TCC_MISS_sum) and memory-side traffic of the grouped backward launches, one PMC group per pass
# over tools/probe_deep_ring.py (CFGS selects the tile configuration).   tools/pmc_l2_bwd.sh [cfg]
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_l2_bwd; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp CFGS=${1:-29} NO_RING=1
i=0
for G in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -- python $ROOT/tools/probe_deep_ring.py > $OUT/g$i.log 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_bf16' not in r['Kernel_Name']: continue
        acc[(r['Grid_Size'], r['Counter_Name'])][r['Dispatch_Id']].append(float(r['Counter_Value']))
res = collections.defaultdict(dict)
for (grid, c), dd in acc.items():
    v = [sum(x) for x in dd.values()]
    v = v[len(v) // 3:]
    res[grid][c] = sum(v) / len(v)
for grid, d in sorted(res.items()):
    hit, miss = d.get('TCC_HIT_sum', 0), d.get('TCC_MISS_sum', 0)
    print(f"grid {grid:>8s}: L2 hit rate {hit / max(1.0, hit + miss):.3f} (hit {hit:.3g}, miss {miss:.3g} requests);  EA read requests {d.get('TCC_EA0_RDREQ_sum', 0):.3g} "
          f"(x 64 B x 2 = {d.get('TCC_EA0_RDREQ_sum', 0) * 128 / 1e6:.1f} MB), write requests {d.get('TCC_EA0_WRREQ_sum', 0):.3g}")
PY
