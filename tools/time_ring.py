"""GPU diagnostic: per-launch time of the persistent ring launch on config 2's backward problems (JAMIE_LIB selects an A/B build:
ablations are timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv, experiments as ex
nv.require_gpu()
B, NBUF = 512, 6
d = tuple(int(v) for v in os.environ.get("DIMS", "2000,1000").split(","))
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def problems(wshape, sks, dx=True):
    probs = []
    if dx:
        for (nout, nin), s1 in zip(wshape, sks):
            probs.append(nv.gemm_problem(T(B, nout), T(nout, nin), torch.empty(s1, B, nin, device='cuda'), B, nin, nout, nout, nin, nin, splitk=s1, slab_stride=B * nin, b_tr=True))
    for (nout, nin) in wshape:
        probs.append(nv.gemm_problem(T(B, nout), T(B, nin), torch.empty(nout, nin, device='cuda', dtype=torch.bfloat16), nout, nin, B, nout, nin, nin,
                                     a_tr=True, b_tr=True, store_nt=True, c_bf16=True))
    return probs
n_wg = torch.cuda.get_device_properties(0).multi_processor_count
err = torch.zeros(4, dtype=torch.int32, device='cuda')
out = []
for name, wshape, sks, dx in (('dec2', [(x, 2 * x) for x in d], (1, 1), True), ('dec1', [(2 * x, x) for x in d], (2, 1), True), ('enc0dW', [(2 * x, x) for x in d], (1, 1), False)):
    sets = [problems(wshape, sks, dx) for _ in range(NBUF)]
    sch = [ex.gemm_bf16_ring_plan(p, n_wg) for p in sets]
    ts = []
    for rep in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(NBUF): ex.gemm_bf16_ring(sets[i], sch[i], n_wg, err)
        ev[0].record()
        for i in range(4 * NBUF): ex.gemm_bf16_ring(sets[i % NBUF], sch[i % NBUF], n_wg, err)
        ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / (4 * NBUF) * 1e3)
    out.append(f'{name} {np.median(ts):.1f}')
print(os.environ.get('JAMIE_LIB', 'product').split('_')[-1], ' | '.join(out), 'us per launch; err', int(err[0]), flush=True)
