"""GPU microbenchmark (round 3): would a BatchNorm-backward launch co-run well with a dW launch?  At config 2, bf16: the
launches one after the other on one stream against the two on two streams at once (no dependency between them in the real
step: dW of layer k reads what BatchNorm backward of layer k already wrote, BatchNorm backward of layer k - 1 reads dX of k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
nv.require_gpu()
dims, L, B = (2000, 1000), 32, 512
torch.manual_seed(0)
model = edModelVar(dims, L)
eng = TrainEngine(model, B, compute_dtype='bf16')
eng.set_batch([torch.randn(B, d).cuda() for d in dims])
eng.forward_backward()
torch.cuda.synchronize()
side = torch.cuda.Stream()


def bn():      # BatchNorm backward of decoder layer 1 (reads the dX slabs of dec2)
    eng._bn_bwd('bn3', 'de2', 'g2', 'dec1', 13, None, 'dec_masks', 1)


def dw():      # dW of dec2 (K = batch)
    eng._dw_gemm('dxhat', 'e2', 'dec2')


def dx():      # dX of dec1
    eng._dx_gemm('de2', 'dec1', 'de1', 'd_e1')


def grouped():
    eng._bwd_gemms('de2', 'dec1', 'e1', 'de1', 'd_e1')


def timeit(fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def both():
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    nv.set_stream(side); dw(); nv.set_stream(None)
    bn()
    ev2 = torch.cuda.Event(); ev2.record(side)
    torch.cuda.current_stream().wait_event(ev2)


eng._fuse_now = False
print('BatchNorm backward alone      %.1f us' % timeit(bn))
print('dW (dec2) alone               %.1f us' % timeit(dw))
print('dX (dec1) alone               %.1f us' % timeit(dx))
print('grouped dW + dX (dec1)        %.1f us' % timeit(grouped))
print('bn then dw, one stream        %.1f us' % timeit(lambda: (bn(), dw())))
print('bn || dw, two streams         %.1f us   (includes two cross-stream events)' % timeit(both))
print('dx then (bn || dw)            %.1f us' % timeit(lambda: (dx(), both())))
print('grouped then bn (as now)      %.1f us' % timeit(lambda: (grouped(), bn())))
