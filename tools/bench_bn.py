"""GPU microbenchmark: BN+LeakyReLU+Dropout kernels at the config-2 shapes (both modalities grouped)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B = 512
def run(Ns, p, bf16, bwd, iters=50):
    keep, probs = [], []
    state = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device='cuda')
    for N in Ns:
        h = torch.randn(1, B, N, device='cuda'); da = torch.randn(1, B, N, device='cuda')
        g, b = torch.ones(N, device='cuda'), torch.zeros(N, device='cuda')
        rm, rv, sm, si = (torch.zeros(N, device='cuda') for _ in range(4)); si += 1
        out = torch.empty(B, N, device='cuda')
        obf = torch.empty(B, N, device='cuda', dtype=torch.bfloat16); oT = torch.empty(N, B, device='cuda', dtype=torch.bfloat16)
        dg, db, dl = (torch.zeros(N, device='cuda') for _ in range(3))
        keep += [h, da, g, b, rm, rv, sm, si, out, obf, oT, dg, db, dl]
        if not bwd:
            pr = nv.BnFwdProblem()
            pr.h, pr.nslab, pr.slab_stride, pr.gamma, pr.beta = nv.ptr(h), 1, B * N, nv.ptr(g), nv.ptr(b)
            pr.running_mean, pr.running_var, pr.save_mean, pr.save_invstd = nv.ptr(rm), nv.ptr(rv), nv.ptr(sm), nv.ptr(si)
            pr.out, pr.mask, pr.B, pr.N, pr.rng_stream = (None if bf16 else nv.ptr(out)), None, B, N, 3
            if bf16: pr.out_bf16, pr.outT_bf16 = nv.ptr(obf), nv.ptr(oT)
        else:
            pr = nv.BnBwdProblem()
            pr.da, pr.nslab, pr.slab_stride = nv.ptr(da), 1, B * N
            pr.h, pr.gamma, pr.beta, pr.save_mean, pr.save_invstd = nv.ptr(h), nv.ptr(g), nv.ptr(b), nv.ptr(sm), nv.ptr(si)
            pr.dgamma, pr.dbeta, pr.dbias_lin, pr.mask = nv.ptr(dg), nv.ptr(db), nv.ptr(dl), None
            pr.B, pr.N, pr.rng_stream, pr.accumulate = B, N, 3, 0
            if bf16: pr.dh_bf16, pr.dhT_bf16, pr.skip_f32 = nv.ptr(obf), nv.ptr(oT), 1
        probs.append(pr)
    f = (lambda: nv.bn_act_bwd(probs, p, state)) if bwd else (lambda: nv.bn_act_fwd(probs, p, state))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for Ns in ((4000, 2000), (2000, 1000)):
    for bwd in (False, True):
        for p in (0.0, 0.6):
            for bf16 in (False, True):
                print(f'N={Ns} {"bwd" if bwd else "fwd"} p={p} bf16_out={bf16}: {run(Ns, p, bf16, bwd):6.1f} us', flush=True)
