#!/bin/bash
# GPU box: PMC passes (counters only, one group per run) over tools/bench_gemm_bf16_one.py; prints per-dispatch means.
# usage: tools/pmc_gemm_bf16.sh "<cfg list>" [shape] [splitk]
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_gemm_${2:-fwd}; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp SHAPE=${2:-fwd} SK=${3:-1}
G1="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
G2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
# (round 3: the derived TA_* / TCP_* groups abort inside rocprofv3 on this image and then hang in its signal handler: dropped;
#  every pass runs under `timeout`)
G6="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum"
for cfg in $1; do
  export CFG=$cfg
  i=0
  for G in "$G1" "$G2" "$G3" "$G6"; do
    i=$((i+1))
    (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/c${cfg}_g$i -- python $ROOT/tools/bench_gemm_bf16_one.py > $OUT/c${cfg}_g$i.log 2>&1)
  done
done
python - <<PY
import csv, glob, collections, os
out = '$OUT'
for d in sorted(glob.glob(out + '/c*_g*')):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm_bf16' not in r['Kernel_Name']: continue
            acc[r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
    tag = os.path.basename(d)
    for c, dd in sorted(acc.items()):
        v = [sum(x) for x in dd.values()]
        v = v[len(v) // 3:]
        print(f'{tag} {c:45s} {sum(v) / len(v):16.1f}  (n={len(v)})')
PY
