"""GPU box, diagnostic build (-DJAMIE_LAT_STAMP): per-workgroup phase timelines of the fused latent kernels inside a real
bf16 training step at config 2 (one plan replay after warm-up), summarised per workgroup class (owner / product chunks).
us since the kernel's first workgroup entered; median and max over the class's workgroups."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar

nv.require_gpu()
dev = torch.device('cuda:0')
dims, L, B, N = (2000, 1000), 32, 512, 20000
torch.manual_seed(666)
model = edModelVar(list(dims), L, device=dev)
eng = TrainEngine(model, B, lr=1e-3, seed=666, compute_dtype='bf16')
data = [torch.randn(N, d, device=dev) for d in dims]
idx = torch.zeros(B, dtype=torch.int32, device=dev)
eng.set_kl_anneal(0.5)
plan = eng.make_plan(data, idx, N, False, None)
for _ in range(20):
    eng.run_plan(plan)
torch.cuda.synchronize()
n_rb = (B + 31) // 32
nblk = 1024
buf = (C.c_ulonglong * (16 * nblk))()
fn = nv.load().jamie_latent_debug_stamps
fn.restype = C.c_int
for rep in range(3):
    eng.run_plan(plan)
    torch.cuda.synchronize()
    assert fn(buf, nblk) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 16).astype(np.float64)
    for name, base, labels in (('fwd', 0, ['entry', 'slabs', 'phaseA+stores', 'partials/-', 'product', 'end']),
                               ('bwd', 8, ['entry', 'owner:stored', 'slabs', 'math', 'staged', 'mfma', 'end(chunk)', 'end(owner)'])):
        s = st[:, base:base + len(labels)]
        live = s[:, 0] > 0
        t0 = s[live, 0].min()
        own = np.arange(nblk) < n_rb
        for cls, sel in (('owner ', live & own), ('chunks', live & ~own)):
            if not sel.any():
                continue
            rel = (s[sel] - t0) / 100.0
            rel[s[sel] == 0] = np.nan
            med = np.nanmedian(rel, axis=0)
            mx = np.nanmax(rel, axis=0)
            print(f'{name} {cls} n={int(sel.sum()):4d}  ' + '  '.join(f'{l} {m:5.1f}/{x:5.1f}' for l, m, x in zip(labels, med, mx)))
    print()
