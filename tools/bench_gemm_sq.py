"""GPU microbenchmark: kernel-core efficiency on large square problems and single (ungrouped) layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
def T(*s): return torch.randn(*s, device='cuda')
def run(layout, M, N, K, cfg, sk=1, iters=10):
    if layout == nv.NT: A, Bm, lda, ldb = T(M, K), T(N, K), K, K
    elif layout == nv.NN: A, Bm, lda, ldb = T(M, K), T(K, N), K, N
    else: A, Bm, lda, ldb = T(K, M), T(K, N), M, N
    Cm = torch.empty(sk, M, N, device='cuda')
    pr = [nv.gemm_problem(A, Bm, Cm, M, N, K, lda, ldb, N, splitk=sk, slab_stride=M * N)]
    for _ in range(2): nv.gemm(pr, layout, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): nv.gemm(pr, layout, cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9
for (M, N, K) in [(4096, 4096, 4096), (2048, 4096, 2048), (512, 4096, 2048), (512, 4000, 2000), (512, 2048, 4096)]:
    for layout, nm in ((nv.NT, 'NT'), (nv.NN, 'NN'), (nv.TN, 'TN')):
        for cfg in (0, 1, 2, 4):
            ms, tf = run(layout, M, N, K, cfg)
            print(f'{nm} {M}x{N}x{K} cfg {cfg}: {ms*1e3:9.1f} us {tf:7.1f} TFLOP/s', flush=True)
