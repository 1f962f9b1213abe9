import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv, experiments as ex
nv.require_gpu()
torch.manual_seed(0)
n_wg = torch.cuda.get_device_properties(0).multi_processor_count
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def run(kind, M, N, K):
    err = torch.zeros(4, dtype=torch.int32, device='cuda')
    if kind == 'dw':
        dy, a = T(K, M), T(K, N)
        o1 = torch.zeros(M, N, device='cuda'); o2 = torch.zeros(M, N, device='cuda')
        mk = lambda o: [nv.gemm_problem(dy, a, o, M, N, K, M, N, N, a_tr=True, b_tr=True)]
        ref = dy.float().t() @ a.float()
    else:
        dy, W = T(M, K), T(K, N)
        o1 = torch.zeros(M, N, device='cuda'); o2 = torch.zeros(M, N, device='cuda')
        mk = lambda o: [nv.gemm_problem(dy, W, o, M, N, K, K, N, N, b_tr=True)]
        ref = dy.float() @ W.float()
    nv.gemm_bf16(mk(o1), 29)
    p2 = mk(o2)
    sc = ex.gemm_bf16_ring_plan(p2, n_wg)
    ex.gemm_bf16_ring(p2, sc, n_wg, err)
    torch.cuda.synchronize()
    bad = (o1 != o2)
    print(f'{kind} M{M} N{N} K{K}: cfg29 vs fp32 matmul max {float((o1 - ref).abs().max()):.3g}; ring mismatches {int(bad.sum())} of {bad.numel()}, err {int(err[0])}')
    if bad.any():
        b = bad.cpu().numpy()
        rows = np.where(b.any(1))[0]; cols = np.where(b.any(0))[0]
        print('   bad rows:', rows[:40], '... n', len(rows)); print('   bad cols:', cols[:40], '... n', len(cols))
        blk = b[:128, :128].reshape(8, 16, 8, 16).any(axis=(1, 3)).astype(int)
        print('   16x16 block map of the first tile (rows = m/16, cols = n/16):'); print(blk)
        # is the ring result a product with some k dropped?  compare with partial sums
        d = (o2 - ref).abs()
        print('   ring vs fp32 matmul max', float(d.max()))
for kind in ('dw', 'dx'):
    run(kind, 128, 128, 64)
    run(kind, 128, 128, 512)
    run(kind, 256, 256, 128)
