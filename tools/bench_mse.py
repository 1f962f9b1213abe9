"""GPU: jamie_mse_cast at config 2's shapes, with / without the fp32 output and the per-tile column sums (round 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B, dims, sks = 512, (2000, 1000), (3, 2)
sets = []
for _ in range(6):
    s = []
    for d, sk in zip(dims, sks):
        s.append(dict(y=torch.randn(sk, B, d, device='cuda'), x=torch.randn(B, d, device='cuda'), d=torch.empty(B, d, device='cuda'),
                      db=torch.empty(B, d, device='cuda', dtype=torch.bfloat16), cp=torch.zeros(8, d, device='cuda'),
                      part=torch.zeros(8 * ((d + 63) // 64), device='cuda')))
    sets.append(s)


def run(with_d, with_cp, n=300):
    def probs(s):
        return [nv.mse_problem(t['y'], t['x'], t['d'] if with_d else None, t['db'], None, partial=t['part'], scale=1e-3, pscale=1e-3,
                               colpart=t['cp'] if with_cp else None) for t in s]
    P = [probs(s) for s in sets]
    for i in range(20):
        nv.mse_cast(P[i % 6])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        nv.mse_cast(P[i % 6])
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for _ in range(2):
    for wd, wc in ((True, False), (False, True), (True, True), (False, False)):
        print(f'fp32 d {wd!s:5}  colpart {wc!s:5}: {run(wd, wc):6.2f} us per launch')
