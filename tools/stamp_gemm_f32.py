"""GPU diagnostic: timelines of the fp32 GEMM launches of config 2 as the engine issues them (forward NT with the planner's
K slices, dX NN, the grouped dW TN launch of 1 / 2 / 4 layers), from in-kernel s_memrealtime stamps (diagnostic build:
tools/stamp_gemm_bf16.sh; JAMIE_LIB=$PWD/tools/libjamie_stamp.so python tools/stamp_gemm_f32.py).  Per workgroup: entry,
tile 0 in LDS, k-loop done, stores retired; per CU: when its last workgroup ended."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv, engine
nv.require_gpu()
lib = nv.load()
B, d = 512, (2000, 1000)
NBUF = 4
if os.environ.get('CFGROWS'):          # tile configuration of the forward / dX launches (17: the mid-k-step barrier loop); CFGDW: dW
    engine.tune(f32_rows_cfg=int(os.environ['CFGROWS']))
R = lambda *s: torch.randn(*s, device='cuda')


def stamps():
    nb = 8192
    buf = (C.c_ulonglong * (8 * nb))()
    fn = lib.jamie_debug_stamps_f32
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, nb) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    a = a[a[:, 3] > 0]
    return a[a[:, 0] > a[:, 0].max() - 60000]          # the last launch only (600 us window)


def report(name, sets, launch, iters=9):
    for i in range(iters):
        launch(sets[i % NBUF])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(sets[iters % NBUF]); e1.record(); torch.cuda.synchronize()
    a = stamps()
    t0 = a[:, 0].min()
    us = lambda x: x / 100.0
    st, pro, loop, epi, end = us(a[:, 0] - t0), us(a[:, 1] - a[:, 0]), us(a[:, 2] - a[:, 1]), us(a[:, 3] - a[:, 2]), us(a[:, 3] - t0)
    nk = a[:, 4] % 1000
    print(f'== {name}: {len(a)} workgroups, event {e0.elapsed_time(e1) * 1e3:.1f} us, last end {end.max():.1f} us')
    q = lambda v: f'min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}'
    print('  start    ', q(st)); print('  prologue ', q(pro)); print('  k-loop   ', q(loop)); print('  ns/k-step', q(1e3 * loop / nk))
    print('  epilogue ', q(epi)); print('  end      ', q(end))
    ghz = (a[:, 7] - a[:, 6]) / ((a[:, 3] - a[:, 0]) * 10.0)
    print(f'  shader clock over each workgroup (s_memtime ticks / s_memrealtime): min {ghz.min():.3f} med {np.median(ghz):.3f} max {ghz.max():.3f} GHz')
    cu = a[:, 5]
    cnt = np.unique(cu, return_counts=True)[1]
    busy, first = {}, {}
    for c, s, e in zip(cu, st, end):
        busy[c] = max(busy.get(c, 0), e)
    bv = np.array(list(busy.values()))
    print(f'  distinct CUs {len(cnt)}, workgroups per CU min {cnt.min()} max {cnt.max()}; per-CU last end: min {bv.min():.1f} med {np.median(bv):.1f} max {bv.max():.1f}')
    # MFMA time a CU needs for its k-steps at 4096 cycles per 128x128x32 k-step and 2.4 GHz
    per_cu = {}
    for c, k in zip(cu, nk):
        per_cu[c] = per_cu.get(c, 0) + k
    pk = np.array(list(per_cu.values())) * 4096 / 2400.0
    print(f'  per-CU matrix-pipe time of its k-steps at 2.4 GHz: min {pk.min():.1f} med {np.median(pk):.1f} max {pk.max():.1f} us')
    # k-loop phases at the middle k-step (waves 0, 5, 10, 15): loop top -> MFMAs issued -> next tile stored -> past the barrier
    kb = (C.c_ulonglong * (16 * 8192))()
    fk = lib.jamie_debug_kstamps_f32
    fk.argtypes = [C.c_void_p, C.c_int]
    assert fk(kb, 8192) == 0
    ks = np.frombuffer(kb, dtype=np.uint64).reshape(8192, 4, 4).astype(np.int64)[:len(a)]
    ok = (ks[:, :, 0] > t0).all(axis=1) & (ks[:, :, 3] >= ks[:, :, 0]).all(axis=1)
    ks = ks[ok] * 10          # ns
    for wv in range(4):
        ph = [np.median(ks[:, wv, j + 1] - ks[:, wv, j]) for j in range(3)]
        print(f'  middle k-step, wave {5 * wv:2d}: top -> MFMAs issued {ph[0]:6.0f} ns, -> tile stored {ph[1]:5.0f} ns, -> past barrier {ph[2]:5.0f} ns;'
              f' top relative to wave 0: {np.median(ks[:, wv, 0] - ks[:, 0, 0]):5.0f} ns, barrier exit {np.median(ks[:, wv, 3] - ks[:, 0, 3]):5.0f} ns')
    if os.environ.get('PERCU'):
        for c in np.unique(cu)[:4]:
            m = cu == c
            o = np.argsort(st[m])
            print('   CU', c, '(nk, start, loop start, loop end, end):',
                  [(int(x), round(float(y), 1), round(float(y + p), 1), round(float(y + p + l), 1), round(float(z), 1))
                   for x, y, p, l, z in zip(nk[m][o], st[m][o], pro[m][o], loop[m][o], end[m][o])])


def fwd(shapes_nk):
    cfg, sks = engine.plan_f32_rows(B, shapes_nk)
    sets = []
    for _ in range(NBUF):
        sets.append([nv.gemm_problem(R(B, K), R(N, K), torch.empty(s, B, N, device='cuda'), B, N, K, K, K, N, splitk=s, slab_stride=B * N)
                     for (N, K), s in zip(shapes_nk, sks)])
    return f'cfg {cfg} slices {sks}', sets, (lambda p: nv.gemm(p, nv.NT, cfg))


def dx(shapes_oi):          # dx [B, in] = dy [B, out] W [out, in]
    cfg, sks = engine.plan_f32_rows(B, [(nin, nout) for (nout, nin) in shapes_oi])
    sets = []
    for _ in range(NBUF):
        sets.append([nv.gemm_problem(R(B, nout), R(nout, nin), torch.empty(s, B, nin, device='cuda'), B, nin, nout, nout, nin, nin,
                                     splitk=s, slab_stride=B * nin) for (nout, nin), s in zip(shapes_oi, sks)])
    return f'cfg {cfg} slices {sks}', sets, (lambda p: nv.gemm(p, nv.NN, cfg))


def dw(layers):             # dW [out, in] = dy [B, out]^T a [B, in]
    sets = []
    for _ in range(NBUF):
        probs = []
        for shapes_oi in layers:
            for (nout, nin) in shapes_oi:
                probs.append(nv.gemm_problem(R(B, nout), R(B, nin), torch.empty(nout, nin, device='cuda'), nout, nin, B, nout, nin, nin,
                                             store_nt=True, partial=torch.empty(4096, device='cuda')))
        sets.append(probs)
    return f'{len(layers)} layers, cfg {int(os.environ.get("CFGDW", engine.F32_CFG_DW_FUSED))}', sets, (lambda p: nv.gemm(p, nv.TN, int(os.environ.get("CFGDW", engine.F32_CFG_DW_FUSED))))


up = [(2 * x, x) for x in d]          # [out, in] of a d -> 2d layer
down = [(x, 2 * x) for x in d]        # 2d -> d
cases = {
    'fwd_d2d': lambda: fwd(up), 'fwd_2dd': lambda: fwd(down),
    'dx_K2d': lambda: dx(up), 'dx_Kd': lambda: dx(down),
    'dw_1': lambda: dw([up]), 'dw_2': lambda: dw([up, down]), 'dw_4': lambda: dw([up, down, up, down]),
}
for k in os.environ.get('CASES', 'fwd_d2d,fwd_2dd,dx_K2d,dx_Kd,dw_1,dw_4').split(','):
    tag, sets, launch = cases[k]()
    report(f'{k} ({tag})', sets, launch)
