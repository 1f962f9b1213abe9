"""GPU microbenchmark: clip+Adam kernel on a 40.3 M-parameter flat buffer (config 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
n = 40345136
p, g, m, v = (torch.randn(n, device='cuda') for _ in range(4)); v = v.abs()
pb = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
hyper = torch.zeros(16); hyper[8:14] = torch.tensor([1e-3, .9, .999, 1e-8, 1.0, 1.0]); hyper = hyper.cuda()
state = torch.tensor([0, 1, 0, 0], dtype=torch.int64, device='cuda')
part = torch.zeros(nv.optim_blocks(n), device='cuda')
for bf in (None, pb):
    for _ in range(3): nv.grad_sqnorm(g, part, state); nv.clip_adam(p, g, m, v, part, hyper, state, bf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): nv.clip_adam(p, g, m, v, part, hyper, state, bf)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    b = (28 + (2 if bf is not None else 0)) * n
    print(f'clip_adam bf16_copy={bf is not None}: {us:7.1f} us  {b/us/1e6:6.2f} TB/s', flush=True)
