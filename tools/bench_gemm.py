"""GPU microbenchmark: fp32 MFMA GEMM tile configurations on the config-2 layer shapes (grouped launches
of both modalities, as the training step issues them).  Prints TFLOP/s per (layout, shape, cfg, splitk)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B = 512
d = (2000, 1000)
dev = 'cuda'
def T(*s): return torch.randn(*s, device=dev)
def run(layout, shapes, cfg, sk, iters=20):
    # shapes: list of (M, N, K)
    probs, keep = [], []
    fl = 0
    for (M, N, K) in shapes:
        if layout == nv.NT: A, Bm = T(M, K), T(N, K); lda, ldb = K, K
        elif layout == nv.NN: A, Bm = T(M, K), T(K, N); lda, ldb = K, N
        else: A, Bm = T(K, M), T(K, N); lda, ldb = M, N
        Cm = torch.empty(sk, M, N, device=dev)
        probs.append(nv.gemm_problem(A, Bm, Cm, M, N, K, lda, ldb, N, splitk=sk, slab_stride=M * N))
        fl += 2.0 * M * N * K
    for _ in range(3): nv.gemm(probs, layout, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): nv.gemm(probs, layout, cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, fl / ms / 1e9
cases = [
    ('NT d->2d ', nv.NT, [(B, 2 * x, x) for x in d]),
    ('NT 2d->d ', nv.NT, [(B, x, 2 * x) for x in d]),
    ('NN dy[2d]W', nv.NN, [(B, x, 2 * x) for x in d]),     # dx[B,d] = dy[B,2d] W[2d,d]
    ('NN dy[d]W ', nv.NN, [(B, 2 * x, x) for x in d]),     # dx[B,2d] = dy[B,d] W[d,2d]
    ('TN dW 2dxd', nv.TN, [(2 * x, x, B) for x in d]),
    ('TN dW dx2d', nv.TN, [(x, 2 * x, B) for x in d]),
]
cfgs = [int(c) for c in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5]
for name, layout, shapes in cases:
    for cfg in cfgs:
        for sk in ((1, 2, 4) if layout != nv.TN else (1,)):
            try:
                ms, tf = run(layout, shapes, cfg, sk)
                print(f'{name} cfg {cfg} splitk {sk}: {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s', flush=True)
            except Exception as e:
                print(f'{name} cfg {cfg} splitk {sk}: ERROR {e}', flush=True)
