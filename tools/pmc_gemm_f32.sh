#!/bin/bash
# GPU box: MFMA-busy and clock counters of the fp32 forward GEMM launches (separate --pmc pass, kernel trace only).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_f32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python $ROOT/tools/bench_gemm_f32_psk.py > $OUT/run.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/a/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gemm_f32' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, 'n', len(next(iter(v.values()))))
PY
