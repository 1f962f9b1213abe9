"""GPU diagnostic: timeline of the float4 BatchNorm forward kernel inside the training step, from in-kernel s_memrealtime stamps
(diagnostic build: jamie_amd.build.build_variant('bnstamp', ['-DJAMIE_BN_STAMP'], only=['bn_act.hip']); run with
JAMIE_LIB=jamie_amd/libjamie_hip_bnstamp.so).  The stamps of the step's LAST forward BatchNorm launch survive (decoder layer 1:
2d-wide, 375 strips, (3, 2) slabs at config 2).  Per workgroup: entry, slabs summed, statistics done, stores issued, stores retired."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
nv.require_gpu()
lib = nv.load()
dims, L, B, N = (2000, 1000), 32, 512, 20000
torch.manual_seed(0)
model = edModelVar(dims, L, device='cuda')
eng = TrainEngine(model, B, compute_dtype='bf16')
data = [torch.randn(N, d, device='cuda') for d in dims]
idx = torch.zeros(B, dtype=torch.int32, device='cuda')
plan = eng.make_plan(data, idx, N, False, None)
for _ in range(20):
    eng.run_plan(plan)
torch.cuda.synchronize()
nb = 4096
buf = (C.c_ulonglong * (8 * nb))()
fn = lib.jamie_debug_bn_stamps
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
a = a[(a[:, 4] > 0) & (a[:, 0] > 0)]
a = a[a[:, 0] > a[:, 0].max() - 10000]                  # the last launch only (100 us window)
t0 = a[:, 0].min()
us = lambda x: x / 100.0
st, ld, stat, sto, ret = us(a[:, 0] - t0), us(a[:, 1] - a[:, 0]), us(a[:, 2] - a[:, 1]), us(a[:, 3] - a[:, 2]), us(a[:, 4] - a[:, 3])
end = us(a[:, 4] - t0)
q = lambda v: f'min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}'
print(f'{len(a)} workgroups of the last forward BatchNorm launch; last end {end.max():.1f} us after the first entry')
print('  entry              ', q(st))
print('  slabs summed       ', q(ld))
print('  statistics         ', q(stat))
print('  normalise + stores ', q(sto))
print('  stores retired     ', q(ret))
print('  end                ', q(end))
for key in sorted(set(a[:, 6])):
    m = a[:, 6] == key
    print(f'  problem N = {key // 10}, {key % 10} slabs: {m.sum()} wgs; entry med {np.median(st[m]):.2f} (max {st[m].max():.2f}), loads {np.median(ld[m]):.2f}, '
          f'stats {np.median(stat[m]):.2f}, stores {np.median(sto[m]):.2f}, retire {np.median(ret[m]):.2f}, whole {np.median(end[m] - st[m]):.2f}, end max {end[m].max():.2f}')
cu = a[:, 5]
cnt = np.unique(cu, return_counts=True)[1]
print(f'  distinct CUs {len(cnt)}, workgroups per CU min {cnt.min()} max {cnt.max()}')
late = st > 3.0
print(f'  workgroups that start more than 3 us after the first: {late.sum()} (entry med {np.median(st[late]) if late.any() else 0:.2f})')
for c in np.unique(cu)[:5]:
    m = cu == c
    o = np.argsort(st[m])
    print('   CU', c, '(start, loaded, stats, stored, retired):', [(round(float(s), 1), round(float(s + l), 1), round(float(s + l + x), 1), round(float(s + l + x + y), 1), round(float(e), 1))
                                                                  for s, l, x, y, e in zip(st[m][o], ld[m][o], stat[m][o], sto[m][o], end[m][o])])
