"""GPU microbenchmark: the four skinny bf16 GEMM launches of the config-2 step (heads, dec0 and their backward)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B, d, L = 512, (2000, 1000), 32
NBUF = int(os.environ.get('NBUF', '6'))
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def run(shapes, sks, cfg, iters=40):
    sets = []
    for b in range(NBUF):
        probs = []
        for (M, N, K), sk in zip(shapes, sks):
            A, Bm = T(M, K), T(N, K)
            Cm = torch.empty(sk, M, N, device='cuda')
            probs.append(nv.gemm_problem(A, Bm, Cm, M, N, K, K, K, N, splitk=sk, slab_stride=M * N))
        sets.append(probs)
    for i in range(3): nv.gemm_bf16(sets[i % NBUF], cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): nv.gemm_bf16(sets[i % NBUF], cfg)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [('head fwd   [B,2L] K=d      ', [(B, 2 * L, x) for x in d], [(1, 1), (3, 3), (7, 7), (15, 7)]),
         ('dec0 fwd   [B,d] K=L       ', [(B, x, L) for x in d], [(1, 1)]),
         ('bwd dec0   dW[d,L]+dX[B,L] ', [(x, L, B) for x in d] + [(B, L, x) for x in d], [(1, 1, 1, 1), (1, 1, 3, 3), (1, 1, 7, 7)]),
         ('bwd head   dW[2L,d]+dX[B,d]', [(2 * L, x, B) for x in d] + [(B, x, 2 * L) for x in d], [(1, 1, 1, 1)])]
for name, shapes, skl in cases:
    for sks in skl:
        for cfg in [int(c) for c in os.environ.get('CFGS', '10,7').split(',')]:
            print(f'{name} cfg {cfg} splitk {sks}: {run(shapes, sks, cfg):7.1f} us', flush=True)
