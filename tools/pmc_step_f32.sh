#!/bin/bash
# GPU box: matrix-pipe and LDS counters of the fp32 step's GEMM launches (separate --pmc passes with the kernel trace only, as
# MI355X_MICROARCH.md prescribes), over a short `bench.py --dtype f32` run.   tools/pmc_step_f32.sh
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_step_f32; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -- python $ROOT/bench.py --dtype f32 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/g$i.log 2>&1)
  echo "pass $i done" >> $OUT/progress.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_f32_kernel<128' not in r['Kernel_Name'] and 'gemm_f32_kernel<256' not in r['Kernel_Name']: continue
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    m = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in d.items()}
    n = len(next(iter(d.values())))
    print(k, ' launches', n)
    print('   ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(m.items())))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        print(f"   matrix pipe busy: {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles per SIMD of {m['GRBM_GUI_ACTIVE'] / 8:.0f} launch cycles per XCD = "
              f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (m['GRBM_GUI_ACTIVE'] / 8):.3f}")
    if 'SQ_WAVE_CYCLES' in m:
        w = m['SQ_WAVE_CYCLES']
        print(f"   wave cycles: waiting to issue {m.get('SQ_WAIT_INST_ANY', 0) / w:.2f}, at waitcnt / barrier {m.get('SQ_WAIT_ANY', 0) / w:.2f}, issuing {m.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f}")
    if 'SQ_LDS_IDX_ACTIVE' in m:
        print(f"   LDS bank-conflict cycles / LDS active cycles = {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m['SQ_LDS_IDX_ACTIVE']):.3f}")
PY
