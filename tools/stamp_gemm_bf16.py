"""GPU diagnostic: timeline of the large-tile bf16 GEMM from in-kernel s_memrealtime stamps (diagnostic build,
tools/stamp_gemm_bf16.sh).  Per workgroup: entry, tile 0 published, k-loop done, stores retired."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv
nv.require_gpu()
lib = nv.load()
B, d = 512, (2000, 1000)
NBUF = int(os.environ.get('NBUF', '6'))
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def run(name, shapes, cfg, sks, iters=13):
    f32 = name.startswith('f32')
    sets = []
    for b in range(NBUF):
        probs = []
        for (M, N, K), s1 in zip(shapes, sks):
            A, Bm = (torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')) if f32 else (T(M, K), T(N, K))
            probs.append(nv.gemm_problem(A, Bm, torch.empty(s1, M, N, device='cuda'), M, N, K, K, K, N, splitk=s1, slab_stride=M * N))
        sets.append(probs)
    launch = (lambda p: nv.gemm(p, nv.NT, cfg)) if f32 else (lambda p: nv.gemm_bf16(p, cfg))
    for i in range(iters): launch(sets[i % NBUF])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(sets[iters % NBUF]); e1.record(); torch.cuda.synchronize()
    nb = 8192
    buf = (C.c_ulonglong * (8 * nb))()
    fn = lib.jamie_debug_stamps_f32 if f32 else lib.jamie_debug_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, nb) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    a = a[a[:, 3] > 0]
    # keep the blocks of the last launch only (stamps of earlier, larger launches would be stale)
    a = a[a[:, 0] > a[:, 0].max() - 30000]          # the last launch only (300 us window)
    t0 = a[:, 0].min()
    us = lambda x: x / 100.0
    st, pro, loop, epi, end = us(a[:, 0] - t0), us(a[:, 1] - a[:, 0]), us(a[:, 2] - a[:, 1]), us(a[:, 3] - a[:, 2]), us(a[:, 3] - t0)
    print(f'== {name} cfg {cfg} sk {sks}: {len(a)} workgroups, event {e0.elapsed_time(e1)*1e3:.1f} us, last end {end.max():.1f} us')
    q = lambda v: f'min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}'
    print('  start    ', q(st)); print('  prologue ', q(pro)); print('  k-loop   ', q(loop)); print('  epilogue ', q(epi)); print('  end      ', q(end))
    if (a[:, 6] > a[:, 0]).all() and (a[:, 7] >= a[:, 6]).all() and (a[:, 1] >= a[:, 7]).all():
        # round 5: entry -> descriptor + addresses ready (6) -> every prologue DMA issued (7) -> tile 0 published (1)
        print('  prologue: entry -> first DMA issue', q(us(a[:, 6] - a[:, 0])))
        print('            issue of the NB tiles   ', q(us(a[:, 7] - a[:, 6])))
        print('            issued -> tile 0 visible', q(us(a[:, 1] - a[:, 7])))
    for pi in sorted(set(a[:, 4] // 1000)):
        m = a[:, 4] // 1000 == pi
        nk = a[m, 4] % 1000
        print(f'  problem {pi}: {m.sum()} wgs, nk {nk.min()}..{nk.max()}, loop med {np.median(loop[m]):.2f} us = {np.median(loop[m]) / np.median(nk) * 1e3:.0f} ns/k-step, epi med {np.median(epi[m]):.2f}, start med {np.median(st[m]):.2f}, end max {end[m].max():.2f}')
    cu = a[:, 5]
    if os.environ.get('PERCU'):
        order = np.argsort(end)
        for c in np.unique(cu)[:6]:
            m = cu == c
            print('   CU', c, 'tiles (problem*1000+nk, start, end):', [(int(x), round(float(y), 1), round(float(z), 1)) for x, y, z in zip(a[m, 4], st[m], end[m])])
    cnt = np.unique(cu, return_counts=True)[1]
    print(f'  distinct CUs {len(cnt)}, workgroups per CU min {cnt.min()} max {cnt.max()}')
    busy = {}
    for c, e in zip(cu, end): busy[c] = max(busy.get(c, 0), e)
    bv = np.array(list(busy.values()))
    print(f'  per-CU last end: min {bv.min():.1f} med {np.median(bv):.1f} max {bv.max():.1f}')
cases = {
 'fwd_d2d': ([(B, 2 * x, x) for x in d], 23, (3, 2)),
 'fwd_2dd': ([(B, x, 2 * x) for x in d], 24, (3, 2)),
 'bwd_dec1': ([(2 * x, x, B) for x in d] + [(B, x, 2 * x) for x in d], 25, (1, 1, 4, 2)),
 'bwd_enc1': ([(x, 2 * x, B) for x in d] + [(B, 2 * x, x) for x in d], 25, (1, 1, 2, 1)),
 'f32_fwd_d2d': ([(B, 2 * x, x) for x in d], -1, (1, 1)),
 'f32_fwd_d2d_sk21': ([(B, 2 * x, x) for x in d], -1, (2, 1)),
 'f32_fwd_2dd': ([(B, x, 2 * x) for x in d], -1, (2, 3)),
 'dw_only': ([(2 * x, x, B) for x in d], 25, (1, 1)),
 'bwd_dec1_r': ([(B, x, 2 * x) for x in d] + [(2 * x, x, B) for x in d], 25, (4, 2, 1, 1)),      # dX problems first
 'bwd_enc1_r': ([(B, 2 * x, x) for x in d] + [(x, 2 * x, B) for x in d], 25, (2, 1, 1, 1)),
}
def run_bwd_now(name, wshape, sks, cfg=29, iters=13):
    """The grouped backward launch as the engine issues it now (engine._bwd_gemms): dX problems first (dy [B, out] x W [out, in]
    as stored, b_tr), then the dW problems (dy^T a on the row-major operands, a_tr + b_tr, bf16 result, nt stores)."""
    sets = []
    for b in range(NBUF):
        probs = []
        for (nout, nin), s1 in zip(wshape, sks):
            dy, W = T(B, nout), T(nout, nin)
            probs.append(nv.gemm_problem(dy, W, torch.empty(s1, B, nin, device='cuda'), B, nin, nout, nout, nin, nin, splitk=s1, slab_stride=B * nin, b_tr=True))
        for (nout, nin) in wshape:
            dy, a = T(B, nout), T(B, nin)
            probs.append(nv.gemm_problem(dy, a, torch.empty(nout, nin, device='cuda', dtype=torch.bfloat16), nout, nin, B, nout, nin, nin,
                                         a_tr=True, b_tr=True, store_nt=True, c_bf16=True))
        sets.append(probs)
    for i in range(iters): nv.gemm_bf16(sets[i % NBUF], cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); nv.gemm_bf16(sets[iters % NBUF], cfg); e1.record(); torch.cuda.synchronize()
    nb = 8192
    buf = (C.c_ulonglong * (8 * nb))()
    fn = lib.jamie_debug_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, nb) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    a = a[a[:, 3] > 0]
    a = a[a[:, 0] > a[:, 0].max() - 30000]
    t0 = a[:, 0].min()
    us = lambda x: x / 100.0
    st, pro, loop, epi, end = us(a[:, 0] - t0), us(a[:, 1] - a[:, 0]), us(a[:, 2] - a[:, 1]), us(a[:, 3] - a[:, 2]), us(a[:, 3] - t0)
    print(f'== {name} cfg {cfg} sk {sks}: {len(a)} workgroups, event {e0.elapsed_time(e1)*1e3:.1f} us, last end {end.max():.1f} us')
    for pi in sorted(set(a[:, 4] // 1000)):
        m = a[:, 4] // 1000 == pi
        nk = a[m, 4] % 1000
        print(f'  problem {pi}: {m.sum()} wgs, nk {nk.min()}..{nk.max()}: start med {np.median(st[m]):.2f} (max {st[m].max():.2f}), prologue med {np.median(pro[m]):.2f}, '
              f'loop med {np.median(loop[m]):.2f} us = {np.median(loop[m]) / np.median(nk) * 1e3:.0f} ns/k-step, epilogue med {np.median(epi[m]):.2f}, whole med {np.median(end[m] - st[m]):.2f}, end max {end[m].max():.2f}')
    cu = a[:, 5]
    for c in np.unique(cu)[:4]:
        m = cu == c
        o = np.argsort(st[m])
        print('   CU', c, '(problem*1000+nk, start, published, loop end, end):',
              [(int(x), round(float(y), 1), round(float(y + p), 1), round(float(y + p + l), 1), round(float(z), 1))
               for x, y, p, l, z in zip(a[m, 4][o], st[m][o], pro[m][o], loop[m][o], end[m][o])])


if os.environ.get('CASES') == 'bwd_now':
    cfgb = int(os.environ.get('CFGBWD', '29'))        # (32: the same tiles on 8 waves with THREE buffers = one workgroup per CU)
    run_bwd_now('bwd dec2 (dX K = d, dW d x 2d)', [(x, 2 * x) for x in d], (1, 1), cfg=cfgb)
    run_bwd_now('bwd dec1 (dX K = 2d, dW 2d x d)', [(2 * x, x) for x in d], (2, 1), cfg=cfgb)
    sys.exit(0)
for k in os.environ.get('CASES', 'fwd_d2d,fwd_2dd,bwd_dec1,bwd_dec1_r,bwd_enc1,bwd_enc1_r').split(','):
    shapes, cfg, sks = cases[k]
    run(k, shapes, int(os.environ.get('CFG', cfg)), sks)
