"""GPU microbenchmark: clip + Adam (fp32 gradient, no bf16 copy: the fp32 compute mode) over flat buffers of several sizes --
config 2 (40 M parameters) to config 5 (233 M) -- TB/s of the 28 algorithmic bytes per parameter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
hyper = torch.zeros(16); hyper[8:14] = torch.tensor([1e-3, .9, .999, 1e-8, 1.0, 1.0]); hyper = hyper.cuda()
state = torch.tensor([0, 1, 0, 0], dtype=torch.int64, device='cuda')
for n in (40345136, 80000000, 120000000, 160000000, 233477632):
    p, g, m, v = (torch.randn(n, device='cuda') for _ in range(4)); v = v.abs()
    part = torch.zeros(nv.optim_blocks(n), device='cuda')
    for _ in range(3): nv.grad_sqnorm(g, part, state); nv.clip_adam(p, g, m, v, part, hyper, state, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): nv.clip_adam(p, g, m, v, part, hyper, state, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f'n = {n:10d}: fp32 gradient {us:8.1f} us  {28 * n / us / 1e6:6.2f} TB/s', end='', flush=True)
    g16, w16 = g.to(torch.bfloat16), torch.zeros(n, device='cuda', dtype=torch.bfloat16)      # bf16 mode: bf16 gradient + bf16 weight copy
    for _ in range(3): nv.clip_adam(p, g16, m, v, part, hyper, state, w16)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): nv.clip_adam(p, g16, m, v, part, hyper, state, w16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f'   bf16 gradient + bf16 weight copy {us:8.1f} us  {28 * n / us / 1e6:6.2f} TB/s', flush=True)
    del p, g, m, v, g16, w16
    torch.cuda.empty_cache()
