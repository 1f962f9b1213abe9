"""GPU microbenchmark: fp32 grouped GEMM launches of config 2 per tile configuration and per-problem split-K
(the data behind engine.plan_f32_rows: 128x128x32 tile, K slices of one common length, <= 2 workgroups per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B, d = 512, (2000, 1000)
NBUF = int(os.environ.get('NBUF', '4'))
def T(*s): return torch.randn(*s, device='cuda')
def run(layout, shapes, sks, cfg, iters=20):
    sets = []
    for _ in range(NBUF):
        probs = []
        for (M, N, K), sk in zip(shapes, sks):
            if layout == nv.NT: A, Bm, lda, ldb = T(M, K), T(N, K), K, K
            elif layout == nv.NN: A, Bm, lda, ldb = T(M, K), T(K, N), K, N
            else: A, Bm, lda, ldb = T(K, M), T(K, N), M, N
            probs.append(nv.gemm_problem(A, Bm, torch.empty(sk, M, N, device='cuda'), M, N, K, lda, ldb, N, splitk=sk, slab_stride=M * N))
        sets.append(probs)
    fl = sum(2.0 * M * N * K for (M, N, K) in shapes)
    for i in range(4): nv.gemm(sets[i % NBUF], layout, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): nv.gemm(sets[i % NBUF], layout, cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, fl / ms / 1e9
if os.environ.get('BIG_TILES') == '1':       # 256x128 / 128x256 on 16 waves (one workgroup per CU) against the 128x128 plan
    tests = [('NT d->2d', nv.NT, [(B, 2 * x, x) for x in d], {12: [(3, 2)], 15: [(3, 2), (2, 2), (4, 2)], 16: [(3, 2)]}),
             ('NT 2d->d', nv.NT, [(B, x, 2 * x) for x in d], {12: [(6, 3)], 15: [(6, 4), (6, 3), (4, 2)], 16: [(6, 4)]}),
             ('NN dy[2d]W', nv.NN, [(B, x, 2 * x) for x in d], {12: [(6, 3)], 15: [(6, 4), (4, 2)], 16: [(6, 4)]}),
             ('NN dy[d]W', nv.NN, [(B, 2 * x, x) for x in d], {12: [(3, 2)], 15: [(3, 2)], 16: [(3, 2)]}),
             ('TN dW 2dxd', nv.TN, [(2 * x, x, B) for x in d], {12: [(1, 1)], 15: [(1, 1)], 16: [(1, 1)]}),
             ('TN dW dx2d', nv.TN, [(x, 2 * x, B) for x in d], {12: [(1, 1)], 15: [(1, 1)], 16: [(1, 1)]})]
else:
    tests = [('NT d->2d', nv.NT, [(B, 2 * x, x) for x in d], {1: [(1, 1)], 4: [(3, 2)], 12: [(3, 2)], 13: [(3, 2)], 14: [(3, 2), (2, 1)]}),
             ('NT 2d->d', nv.NT, [(B, x, 2 * x) for x in d], {1: [(4, 2)], 4: [(6, 3)], 12: [(6, 3), (6, 4)], 13: [(6, 3)], 14: [(6, 3), (3, 2)]}),
             ('NN dy[2d]W', nv.NN, [(B, x, 2 * x) for x in d], {1: [(4, 2)], 4: [(6, 3)], 12: [(6, 3), (6, 4)]}),
             ('NN dy[d]W', nv.NN, [(B, 2 * x, x) for x in d], {1: [(1, 1)], 4: [(3, 2)], 12: [(3, 2)]}),
             ('TN dW 2dxd', nv.TN, [(2 * x, x, B) for x in d], {1: [(1, 1)], 12: [(1, 1)], 4: [(1, 1)]}),
             ('TN dW dx2d', nv.TN, [(x, 2 * x, B) for x in d], {1: [(1, 1)], 12: [(1, 1)]})]
for name, layout, shapes, plan in tests:
    for cfg, skl in plan.items():
        for sks in skl:
            us, tf = run(layout, shapes, sks, cfg)
            print(f'{name:12s} cfg {cfg} sk {sks}: {us:8.1f} us {tf:7.1f} TFLOP/s', flush=True)
