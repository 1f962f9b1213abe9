"""GPU helper for PMC passes: one grouped bf16 GEMM launch shape, repeated (CFG, SHAPE = fwd|fwd2|dw, SK env)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
cfg = int(os.environ.get('CFG', '7'))
sks = [int(v) for v in os.environ.get('SK', '1').split(',')]          # per problem (the last value repeats): SK=3,2
B, d = 512, (2000, 1000)
shapes = {'fwd': [(B, 2 * x, x) for x in d], 'fwd2': [(B, x, 2 * x) for x in d], 'dw': [(2 * x, x, B) for x in d]}[os.environ.get('SHAPE', 'fwd')]
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
sets = []
for b in range(12):
    probs = []
    for j, (M, N, K) in enumerate(shapes):
        sk = sks[min(j, len(sks) - 1)]
        A, Bm = T(M, K), T(N, K)
        Cm = torch.empty(sk, M, N, device='cuda')
        probs.append(nv.gemm_problem(A, Bm, Cm, M, N, K, K, K, N, splitk=sk, slab_stride=M * N))
    sets.append(probs)
for i in range(36): nv.gemm_bf16(sets[i % 12], cfg)
torch.cuda.synchronize()
