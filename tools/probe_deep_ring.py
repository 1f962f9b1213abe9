"""GPU diagnostic: the grouped backward launch (dX on W as stored + dW on the activations as stored) with 2 .. 5 LDS buffers
per 128 x 128 tile (configurations 29 / 32 / 33 / 34: one, two, three, four k-tiles in flight per workgroup): results must be
bit-identical; per-launch time by HIP events over rotating operand sets (cold operands, as in the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv, experiments as ex
nv.require_gpu()
B = 512
d = tuple(int(v) for v in os.environ.get('DIMS', '2000,1000').split(','))
NBUF = 6
def T(*s): return torch.randn(*s, device='cuda').to(torch.bfloat16)
def make(wshape, sks):
    sets = []
    for b in range(NBUF):
        probs, outs = [], []
        for (nout, nin), s1 in zip(wshape, sks):
            dy, W = T(B, nout), T(nout, nin)
            o = torch.zeros(s1, B, nin, device='cuda')
            outs.append(o)
            probs.append(nv.gemm_problem(dy, W, o, B, nin, nout, nout, nin, nin, splitk=s1, slab_stride=B * nin, b_tr=True))
        for (nout, nin) in wshape:
            dy, a = T(B, nout), T(B, nin)
            o = torch.zeros(nout, nin, device='cuda', dtype=torch.bfloat16)
            outs.append(o)
            probs.append(nv.gemm_problem(dy, a, o, nout, nin, B, nout, nin, nin, a_tr=True, b_tr=True, store_nt=True, c_bf16=True))
        sets.append((probs, outs))
    return sets
for name, wshape, sks in (('dec2', [(x, 2 * x) for x in d], (1, 1)), ('dec1', [(2 * x, x) for x in d], (2, 1)),
                          ('enc0 dW only', None, None)):
    if wshape is None:
        sets = []
        for b in range(NBUF):
            probs, outs = [], []
            for x in d:
                dy, a = T(B, 2 * x), T(B, x)
                o = torch.zeros(2 * x, x, device='cuda', dtype=torch.bfloat16)
                outs.append(o)
                probs.append(nv.gemm_problem(dy, a, o, 2 * x, x, B, 2 * x, x, x, a_tr=True, b_tr=True, store_nt=True, c_bf16=True))
            sets.append((probs, outs))
    else:
        sets = make(wshape, sks)
    ref = None
    for cfg in [int(v) for v in os.environ.get('CFGS', '29,32,33,34').split(',')]:
        nv.gemm_bf16(sets[0][0], cfg)
        torch.cuda.synchronize()
        got = [o.clone() for o in sets[0][1]]
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), (name, cfg)
        ts = []
        for rep in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for i in range(NBUF): nv.gemm_bf16(sets[i][0], cfg)
            ev[0].record()
            for i in range(4 * NBUF): nv.gemm_bf16(sets[i % NBUF][0], cfg)
            ev[1].record(); torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) / (4 * NBUF) * 1e3)
        print(f'{name}: cfg {cfg}  {np.median(ts):7.1f} us per launch (min {min(ts):.1f})  == cfg 29: ok', flush=True)
    if os.environ.get('NO_RING') == '1' or not ex.available():
        continue
    # the persistent loader / consumer ring launch (jamie_gemm_bf16_ring) on the same problems
    n_wg = torch.cuda.get_device_properties(0).multi_processor_count
    err = torch.zeros(4, dtype=torch.int32, device='cuda')
    scheds = [ex.gemm_bf16_ring_plan(st[0], n_wg) for st in sets]
    assert all(sc is not None for sc in scheds)
    for o in sets[0][1]: o.zero_()
    ex.gemm_bf16_ring(sets[0][0], scheds[0], n_wg, err)
    torch.cuda.synchronize()
    assert int(err[0].item()) == 0, ('ring hand-off error word', int(err[0].item()))
    bad = [i for i, (a, b) in enumerate(zip(ref, sets[0][1])) if not torch.equal(a, b)]
    if bad:
        for i in bad:
            a, b = ref[i].float(), sets[0][1][i].float()
            print(f'  MISMATCH output {i}: {float((a - b).abs().max()):.4g} max abs, {float((a != b).float().mean()):.4g} of the elements', flush=True)
    ts = []
    for rep in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(NBUF): ex.gemm_bf16_ring(sets[i][0], scheds[i], n_wg, err)
        ev[0].record()
        for i in range(4 * NBUF): ex.gemm_bf16_ring(sets[i % NBUF][0], scheds[i % NBUF], n_wg, err)
        ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / (4 * NBUF) * 1e3)
    print(f'{name}: RING    {np.median(ts):7.1f} us per launch (min {min(ts):.1f})  == cfg 29: {"ok" if not bad else "MISMATCH"}  err word {int(err[0].item())}', flush=True)
