"""GPU diagnostic: in-kernel timeline of the fused Linear + BatchNorm launch (jamie_gemm_bf16_bn) from s_memrealtime stamps
(diagnostic build: tools/stamp_gemm_bf16.sh; JAMIE_HIP_LIB=$PWD/tools/libjamie_stamp.so python tools/stamp_gemm_bf16_bn.py).
Per workgroup: entry, tile 0 published, k-loop done, slab stores drained, ticket / wait done, BatchNorm strips done."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv, experiments as ex
nv.require_gpu()
lib = nv.load()
B, d = 512, (2000, 1000)
NBUF = 4


def T(*s):
    return torch.randn(*s, device='cuda').to(torch.bfloat16)


def build(shapes, sks):
    sets = []
    for _ in range(NBUF):
        gp, bp, keep = [], [], []
        for i, ((M, N, K), sk) in enumerate(zip(shapes, sks)):
            A, W, h = T(M, K), T(N, K) * K ** -0.5, torch.empty(sk, M, N, device='cuda')
            bias, ga, be = torch.randn(N, device='cuda'), torch.rand(N, device='cuda') + .5, torch.randn(N, device='cuda')
            rm, rv, sm, si = (torch.zeros(N, device='cuda') for _ in range(4))
            out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
            gp.append(nv.gemm_problem(A, W, h, M, N, K, K, K, N, bias=bias, splitk=sk, slab_stride=M * N))
            pr = nv.BnFwdProblem()
            pr.h, pr.nslab, pr.slab_stride = nv.ptr(h), sk, M * N
            pr.gamma, pr.beta, pr.running_mean, pr.running_var = nv.ptr(ga), nv.ptr(be), nv.ptr(rm), nv.ptr(rv)
            pr.save_mean, pr.save_invstd, pr.out, pr.mask, pr.out_bf16 = nv.ptr(sm), nv.ptr(si), None, None, nv.ptr(out)
            pr.B, pr.N, pr.rng_stream = M, N, 10 + 8 * i
            bp.append(pr)
            keep += [A, W, h, bias, ga, be, rm, rv, sm, si, out]
        sets.append((gp, bp, keep))
    return sets


def run(name, shapes, cfg, sks, mode, iters=9):
    sets = build(shapes, sks)
    state = torch.tensor([7, 3, 0, 0], dtype=torch.int64, device='cuda')
    tickets = torch.zeros(4 + 2 * sum((N + 127) // 128 for _, N, _ in shapes), dtype=torch.int32, device='cuda')
    if mode == 0:
        launch = lambda s: (nv.gemm_bf16(s[0], cfg), nv.bn_act_fwd(s[1], 0.6, state))      # noqa: E731
    else:
        launch = lambda s: ex.gemm_bf16_bn(s[0], s[1], cfg, 0.6, state, tickets, mode)     # noqa: E731
    for i in range(iters):
        launch(sets[i % NBUF])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(sets[iters % NBUF]); e1.record(); torch.cuda.synchronize()
    nb = 8192
    buf = (C.c_ulonglong * (8 * nb))()
    fn = lib.jamie_debug_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, nb) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    a = a[a[:, 3] > 0]
    a = a[a[:, 0] > a[:, 0].max() - 30000]
    t0 = a[:, 0].min()
    us = lambda x: x / 100.0      # noqa: E731
    q = lambda v: f'min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}  (n {len(v)})'   # noqa: E731
    print(f'== {name} cfg {cfg} sk {sks} mode {mode}: {len(a)} workgroups, event {e0.elapsed_time(e1) * 1e3:.1f} us')
    print('  entry            ', q(us(a[:, 0] - t0)))
    print('  tile 0 published ', q(us(a[:, 1] - t0)))
    print('  k-loop done      ', q(us(a[:, 2] - t0)))
    print('  stores drained   ', q(us(a[:, 3] - t0)), ' phase', q(us(a[:, 3] - a[:, 2])))
    if mode:
        m = a[:, 6] > a[:, 3]
        print('  wait done        ', q(us(a[m, 6] - t0)), ' phase', q(us(a[m, 6] - a[m, 3])))
        m = a[:, 7] > a[:, 3]
        r = m & (a[:, 7] - a[:, 6] > 50)
        print('  strips done      ', q(us(a[m, 7] - t0)), ' phase (reducers)', q(us(a[r, 7] - a[r, 6])) if r.any() else '')


shapes = [(B, 2 * x, x) for x in d]
for mode in (0, 1, 2):
    run('fwd_d2d', shapes, 31, (3, 2), mode)
shapes = [(B, x, 2 * x) for x in d]
for mode in (0, 1, 2):
    run('fwd_2dd', shapes, 32, (3, 2), mode)
