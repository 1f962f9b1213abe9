"""GPU diagnostic: per-tile timeline of the persistent ring launch (jamie_gemm_bf16_ring) from in-kernel s_memrealtime stamps
(diagnostic build: tools/stamp_gemm_bf16.sh, JAMIE_HIP_LIB=tools/libjamie_stamp.so): consumer wave 0 of every workgroup stamps the
end of each tile's k-loop and of its stores, loader wave 0 its start and end."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv, experiments as ex
nv.require_gpu()
lib = nv.load()
B, d = 512, (2000, 1000)
NBUF = int(os.environ.get('NBUF', '6'))
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def problems(wshape, sks, dx=True):
    probs = []
    if dx:
        for (nout, nin), s1 in zip(wshape, sks):
            dy, W = T(B, nout), T(nout, nin)
            probs.append(nv.gemm_problem(dy, W, torch.empty(s1, B, nin, device='cuda'), B, nin, nout, nout, nin, nin, splitk=s1, slab_stride=B * nin, b_tr=True))
    for (nout, nin) in wshape:
        dy, a = T(B, nout), T(B, nin)
        probs.append(nv.gemm_problem(dy, a, torch.empty(nout, nin, device='cuda', dtype=torch.bfloat16), nout, nin, B, nout, nin, nin,
                                     a_tr=True, b_tr=True, store_nt=True, c_bf16=True))
    return probs
n_wg = torch.cuda.get_device_properties(0).multi_processor_count
err = torch.zeros(4, dtype=torch.int32, device='cuda')
for name, wshape, sks, dx in (('dec2', [(x, 2 * x) for x in d], (1, 1), True), ('enc0 dW only', [(2 * x, x) for x in d], (1, 1), False)):
    sets = [problems(wshape, sks, dx) for _ in range(NBUF)]
    sch = [ex.gemm_bf16_ring_plan(p, n_wg) for p in sets]
    for i in range(13): ex.gemm_bf16_ring(sets[i % NBUF], sch[i % NBUF], n_wg, err)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ex.gemm_bf16_ring(sets[13 % NBUF], sch[13 % NBUF], n_wg, err); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 512))()
    fn = lib.jamie_debug_ring_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, 512) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 64).astype(np.int64)[:n_wg]
    t0 = a[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    sc = sch[13 % NBUF].cpu().numpy().reshape(n_wg, -1)
    print(f'== {name}: event {e0.elapsed_time(e1) * 1e3:.1f} us; consumer end med {np.median(us(a[:, 62])):.1f} max {us(a[:, 62]).max():.1f}; loader end med {np.median(us(a[:, 63])):.1f}; err {int(err[0])}')
    for w in (0, 1, 8, 100, 255):
        n = int((sc[w] >= 0).sum())
        kinds = [f'p{int(c) >> 24}' for c in sc[w][:n]]
        tl = [(kinds[i], round(float(us(a[w, 2 + 2 * i])), 1), round(float(us(a[w, 3 + 2 * i])), 1)) for i in range(min(n, 30))]
        print(f'   wg {w}: start {us(a[w, 0]):.1f}; (problem, k-loop done, stores issued): {tl}; end {us(a[w, 62]):.1f}')
