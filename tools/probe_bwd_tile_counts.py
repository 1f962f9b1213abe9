"""GPU diagnostic: would moving weight-gradient tiles between the last two grouped backward launches of the bf16 step pay?  The enc0
dW launch has 640 tiles of 128 x 128 (K = batch = 512) = 1.25 rounds of the chip's 512 workgroup slots; the enc1 launch in front
of it carries 192 long dX tiles + 640 dW tiles.  Timed with rotating operand sets: dW launches of 512 ... 1024 tiles, and the enc1
launch with 640 / 384 / 256 dW tiles beside its dX problems.  (Shapes with the same per-tile work; synthetic operands.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv
nv.require_gpu()
B, NBUF, CFG = 512, 6, 29
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def dw(nout, nin):
    return nv.gemm_problem(T(B, nout), T(B, nin), torch.empty(nout, nin, device='cuda', dtype=torch.bfloat16), nout, nin, B, nout, nin, nin,
                           a_tr=True, b_tr=True, store_nt=True, c_bf16=True)
def dx(nout, nin, sk):
    return nv.gemm_problem(T(B, nout), T(nout, nin), torch.empty(sk, B, nin, device='cuda'), B, nin, nout, nout, nin, nin, splitk=sk,
                           slab_stride=B * nin, b_tr=True)
def tiles(shapes): return sum(-(-a // 128) * -(-b // 128) for a, b in shapes)
def run(name, make):
    sets = [make() for _ in range(NBUF)]
    ts = []
    for rep in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(NBUF): nv.gemm_bf16(sets[i], CFG)
        ev[0].record()
        for i in range(4 * NBUF): nv.gemm_bf16(sets[i % NBUF], CFG)
        ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / (4 * NBUF) * 1e3)
    print(f'{name:70s} {np.median(ts):6.1f} us per launch', flush=True)
    return float(np.median(ts))
dws = {512: [(4000, 2000)], 640: [(4000, 2000), (2000, 1000)], 768: [(4000, 2000), (2000, 1000), (2048, 1024)], 896: [(4000, 2000), (2000, 1000), (2048, 2048)],
       1024: [(4000, 2000), (2000, 1000), (3072, 2048)], 1280: [(4000, 2000), (2000, 1000), (4000, 2000), (2000, 1000)]}
last = {n: run(f'dW only, {tiles(s)} tiles', lambda s=s: [dw(*x) for x in s]) for n, s in dws.items()}
enc1_dx = [(2000, 4000, 1), (1000, 2000, 1)]        # dx [B, 2d] = dy [B, d] W [d, 2d]: K = d
front = {}
for n, s in ((640, [(2000, 4000), (1000, 2000)]), (512, [(2000, 4000)]), (384, [(2048, 3072)]), (256, [(2048, 2048)]), (128, [(1000, 2000)]), (0, [])):
    front[n] = run(f'enc1: dX (192 long tiles) + {tiles(s)} dW tiles', lambda s=s: [dx(*x) for x in enc1_dx] + [dw(*x) for x in s])
print('sum now (640 + 640):', round(front[640] + last[640], 1))
for moved in (128, 256, 384, 640):
    if 640 - moved in front and 640 + moved in last:
        print(f'move {moved} tiles to the last launch:', round(front[640 - moved] + last[640 + moved], 1))
