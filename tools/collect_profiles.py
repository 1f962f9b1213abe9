#!/usr/bin/env python
"""Turn the outputs of tools/profile_bench.sh (gpurun_out/<tag>/) into the committed summaries under
profiles/: kernel stats CSV, bench JSON and HBM traffic of the dominant kernel (profiles/traffic.json).

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes, so it
is doubled (MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact for 16-byte-per-lane / dword-per-lane stores."""
import csv, glob, json, os, sys
tag, name = sys.argv[1], sys.argv[2]          # e.g. prof3 r01_c2_f32_v3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag)
dst = os.path.join(root, 'profiles')
bench = json.load(open(os.path.join(src, 'bench.json')))
KERNEL = bench['roofline']['kernel'].split(' (')[0] if bench['dtype'] == 'f32' else 'clip_adam_kernel'
def avg_counter(d, counter):
    f = max(glob.glob(os.path.join(src, d, '*', '*_counter_collection.csv')), key=os.path.getmtime)      # (the newest: gpurun_out/ keeps earlier runs)
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if r['Counter_Name'] == counter and KERNEL in r['Kernel_Name']]
    return sum(vals) / len(vals), len(vals)
st = max(glob.glob(os.path.join(src, 'stats', '*', '*_kernel_stats.csv')), key=os.path.getmtime)
rows = list(csv.DictReader(open(st)))
with open(os.path.join(dst, name + '_kernel_stats.csv'), 'w') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
fetch, n1 = avg_counter('pmc_fetch', 'FETCH_SIZE')
write, n2 = avg_counter('pmc_write', 'WRITE_SIZE')
hbm = (2 * fetch + write) * 1024
tr_path = os.path.join(dst, 'traffic.json')
tr = json.load(open(tr_path)) if os.path.exists(tr_path) else {}
cfg = bench['config']['workload'].split(':')[0] + '_' + bench['dtype']
tr[cfg] = {'kernel': KERNEL, 'hbm_bytes_per_launch': hbm, 'fetch_size_kib_raw': fetch, 'write_size_kib_raw': write,
           'launches_averaged': [n1, n2], 'correction': 'FETCH_SIZE x2 (gfx950), KiB -> bytes', 'source': name}
# the kernel north_star names -- the encoder's first Linear, forward d -> 2d (reference model.py:151) -- gets an entry of its
# own (bench.py: roofline.encoder_gemm.hbm_bytes): bf16: the 256 x 128-tile instance that only the d -> 2d launches use (enc0 and
# dec1, identical shapes); fp32: the one kernel instance every forward / dX launch of the large layers shares (averaged over them)
ENC = 'gemm_bf16_dma2_kernel<256, 128, 4, 4, 1, 3, 0>' if bench['dtype'] == 'bf16' else KERNEL


def avg_counter_of(kernel, d, counter):
    f = max(glob.glob(os.path.join(src, d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and kernel in r['Kernel_Name']]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


ef, en1 = avg_counter_of(ENC, 'pmc_fetch', 'FETCH_SIZE')
ew, en2 = avg_counter_of(ENC, 'pmc_write', 'WRITE_SIZE')
ek = [r for r in rows if ENC in r['Name']]
if ef is not None and ew is not None:
    tr[cfg + '_encoder_gemm'] = {'kernel': ENC, 'hbm_bytes_per_launch': (2 * ef + ew) * 1024, 'fetch_size_kib_raw': ef, 'write_size_kib_raw': ew,
                                 'launches_averaged': [en1, en2], 'correction': 'FETCH_SIZE x2 (gfx950), KiB -> bytes',
                                 'rocprofv3_avg_launch_us': float(ek[0]['AverageNs']) / 1e3 if ek else None,
                                 'note': 'bf16: the d -> 2d forward launches (enc0 and dec1: identical shapes); fp32: averaged over every forward / dX launch of the large layers',
                                 'source': name}
json.dump(tr, open(tr_path, 'w'), indent=1)
bench['roofline']['traffic'] = hbm
# the same command under rocprofv3 (--kernel-trace --stats): its own event timing next to the profiler's average, so the
# two clocks can be compared on the SAME run (profiled runs are slower than the un-profiled bench, MI355X_MICROARCH.md DVFS 2)
try:
    prof = json.loads(open(os.path.join(src, 'stats.json')).read().strip().splitlines()[-1])
    kk = [r for r in rows if KERNEL in r['Name']]
    bench['profiled_run'] = {'value': prof['value'], 'ms_per_step': prof['ms_per_step'],
                             'kernel_event_timing_ms': prof['kernel_event_timing_ms'],
                             'rocprofv3_avg_us': float(kk[0]['AverageNs']) / 1e3 if kk else None, 'kernel': KERNEL}
except Exception as e:      # noqa: BLE001
    bench['profiled_run'] = {'error': str(e)}
json.dump(bench, open(os.path.join(dst, name + '_bench.json'), 'w'), indent=1)
k = [r for r in rows if KERNEL in r['Name']]
print('bench ms/step', bench['ms_per_step'], 'cells/s', bench['value'])
print('rocprof avg us for kernel:', float(k[0]['AverageNs']) / 1e3 if k else None, 'bench avg_launch_ms', bench['roofline']['avg_launch_ms'])
print('traffic bytes/launch', hbm, 'algorithmic bytes/launch', bench['roofline'].get('bytes_per_launch', 4 * sum(2 * d * d for d in bench['config']['features'])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:86]:86s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):6.2f}%")
