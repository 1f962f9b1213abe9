"""GPU microbenchmark: does the relative placement of the four streams of clip + Adam (p, g, m, v; + the bf16 weight copy) matter?
One pool, the arrays carved out at chosen byte offsets from 2 MB-aligned bases; several repetitions per placement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
n = int(os.environ.get("N", "40345136"))
hyper = torch.zeros(16); hyper[8:14] = torch.tensor([1e-3, .9, .999, 1e-8, 1.0, 1.0]); hyper = hyper.cuda()
state = torch.tensor([0, 1, 0, 0], dtype=torch.int64, device='cuda')
part = torch.zeros(nv.optim_blocks(n), device='cuda')
SL = (n * 4 + (8 << 20)) // (2 << 20) * (2 << 20) + (2 << 20)       # slot per array: 2 MB multiple with slack
pool = torch.zeros(6 * SL // 4, device='cuda', dtype=torch.float32)
base = pool.data_ptr()
assert base % (2 << 20) == 0 or True
print('pool base mod 2MB', base % (2 << 20))


def carve(slot, off_bytes, dtype=torch.float32, count=n):
    start = (slot * SL + off_bytes) // 4
    t = pool[start:start + (count if dtype == torch.float32 else count // 2)]
    return t if dtype == torch.float32 else t.view(torch.bfloat16)[:count]


def run(offs, g16):
    p, m, v = carve(0, offs[0]), carve(1, offs[1]), carve(2, offs[2])
    g = carve(3, offs[3], torch.bfloat16 if g16 else torch.float32)
    pb = carve(4, offs[4], torch.bfloat16)
    p.normal_(); m.zero_(); v.fill_(1e-3)
    if g16: g.copy_(torch.randn(n, device='cuda').to(torch.bfloat16))
    else: g.normal_()
    nv.grad_sqnorm(g, part, state)
    for _ in range(3): nv.clip_adam(p, g, m, v, part, hyper, state, pb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): nv.clip_adam(p, g, m, v, part, hyper, state, pb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


K = 1024
placements = {'all aligned': (0, 0, 0, 0, 0), '256 B steps': (0, 256, 512, 768, 1024), '1 KB steps': (0, K, 2 * K, 3 * K, 4 * K),
              '4 KB steps': (0, 4 * K, 8 * K, 12 * K, 16 * K), '64 KB steps': (0, 64 * K, 128 * K, 192 * K, 256 * K),
              '512 KB steps': (0, 512 * K, 1024 * K, 1536 * K, 2048 * K), 'odd mix': (0, 4352, 70 * K + 256, 1300 * K + 512, 33 * K)}
for g16 in ((True,) if os.environ.get("G16_ONLY") else (True, False)):
    for rep in range(2):
        for name, offs in placements.items():
            print(f'g16={int(g16)} {name:14s} {run(offs, g16):7.1f} us', flush=True)
