import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
from jamie_amd import _native as nv
nv.require_gpu()
torch.manual_seed(0)
for cfg, (M, N, K) in ((24, (128, 128, 64)), (23, (512, 256, 128)), (26, (128, 72, 200))):
    a = torch.randn(M, K).to(torch.bfloat16); w = torch.randn(N, K).to(torch.bfloat16)
    for rep in range(3):
        out = torch.full((M, N), float('nan'), device='cuda')
        nv.gemm_bf16([nv.gemm_problem(a.cuda(), w.cuda(), out, M, N, K, K, K, N)], cfg)
        torch.cuda.synchronize()
        ref = a.double() @ w.double().t()
        err = (out.cpu().double() - ref).abs()
        bad = (err > 1e-2) | torch.isnan(out.cpu())
        idx = bad.nonzero()
        print(cfg, (M, N, K), 'rep', rep, 'bad', int(bad.sum()), 'nan', int(torch.isnan(out).sum()))
        if len(idx):
            rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
            print('   rows', rows[:40], '... cols', cols[:40])
