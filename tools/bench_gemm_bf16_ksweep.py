"""GPU microbenchmark: fixed cost vs per-k-step cost of the bf16 GEMM (M = 512, N = 4000 + 2000 grouped, K swept)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
NBUF = int(os.environ.get('NBUF', '12'))
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def run(shapes, cfg, sk, iters=48):
    sets = []
    for b in range(NBUF):
        probs = []
        for (M, N, K) in shapes:
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
for cfg in [int(c) for c in os.environ.get('CFGS', '7,14').split(',')]:
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        us = run([(512, 4000, K), (512, 2000, K)], cfg, 1)
        print(f'cfg {cfg} K {K:5d}: {us:7.1f} us   ({us / (K / 64):6.2f} us per k-step)', flush=True)
