"""GPU microbenchmark: bf16 MFMA GEMM (NT, both operands K-contiguous) on the config-2 layer shapes, grouped."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B, d = 512, (2000, 1000)
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
NBUF = int(os.environ.get('NBUF', '1'))     # > 1: rotate through distinct operand sets (cold L2 / Infinity Cache)
def run(shapes, cfg, sk, iters=24):
    sets, fl = [], 0
    sks = sk if isinstance(sk, (list, tuple)) else [sk] * len(shapes)
    for b in range(NBUF):
        probs = []
        for (M, N, K), s1 in zip(shapes, sks):
            A, Bm = T(M, K), T(N, K)
            Cm = torch.empty(s1, M, N, device='cuda')
            probs.append(nv.gemm_problem(A, Bm, Cm, M, N, K, K, K, N, splitk=s1, slab_stride=M * N))
        sets.append(probs)
    fl = sum(2.0 * M * N * K for (M, N, K) in shapes)
    for i in range(3): nv.gemm_bf16(sets[i % NBUF], cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): nv.gemm_bf16(sets[i % NBUF], cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, fl / ms / 1e9
cases = [('fwd d->2d ', [(B, 2 * x, x) for x in d]), ('fwd 2d->d ', [(B, x, 2 * x) for x in d]),
         ('dW  2dxd  ', [(2 * x, x, B) for x in d]), ('dW  dx2d  ', [(x, 2 * x, B) for x in d]),
         ('dX  2d->d ', [(B, x, 2 * x) for x in d]),
         ('bwd enc1  ', [(x, 2 * x, B) for x in d] + [(B, 2 * x, x) for x in d]),      # dW [d,2d] + dX [B,2d] K=d
         ('bwd dec1  ', [(2 * x, x, B) for x in d] + [(B, x, 2 * x) for x in d])]      # dW [2d,d] + dX [B,d] K=2d
if 'CASES' in os.environ:
    cases = [c for c in cases if any(k in c[0] for k in os.environ['CASES'].split(','))]
for name, shapes in cases:
    for cfg in [int(c) for c in os.environ.get('CFGS', '1').split(',')]:
        for sk in ([tuple(int(x) for x in p.split(',')) for p in os.environ['PSK'].split(';')] if 'PSK' in os.environ else (1, 2, 4)):
            if 'dW' in name and sk != 1 and 'PSK' not in os.environ: continue
            ms, tf = run(shapes, cfg, sk)
            print(f'{name} cfg {cfg} splitk {sk}: {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s', flush=True)
