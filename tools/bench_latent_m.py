"""GPU microbenchmark of the fused latent kernels (jamie_latent_m_fwd / _bwd) at config 2's shape, with parts switched off
through the descriptor (VARIANT env: full | nodec (no decoder product) | eps (explicit eps: no Philox) | nobf)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
M, B, L, dims, nslab = 2, 512, int(os.environ.get('L', '32')), (2000, 1000), 3
f32 = dict(device='cuda', dtype=torch.float32)
keep = {}
def T(*s): return torch.randn(*s, **f32)
def desc(variant):
    d = nv.LatentM()
    d.B, d.L, d.M = B, L, M
    k = keep.setdefault(variant, {})
    k.update(sigma=torch.rand(M, **f32) + .2, hyper=torch.ones(16, **f32), rec=torch.rand(5, **f32), losses=torch.zeros(8, **f32),
             partials=torch.zeros(20 * nv.load().jamie_max_partials(), **f32), dsigma=torch.zeros(M, **f32), comb=T(B, L),
             colpart=torch.zeros(int(nv.load().jamie_latent_m_colpart_size(B, L)), **f32))
    for i in range(M):
        k[f'ml{i}'], k[f'hb{i}'], k[f'eps_in{i}'] = T(nslab, B, 2 * L), T(2 * L), T(B, L)
        k[f'W{i}'], k[f'b{i}'], k[f'dcomb{i}'] = T(dims[i], L), T(dims[i]), T(2, B, L)
        k[f'g1{i}'], k[f'dml{i}'], k[f'db{i}'], k[f'alias{i}'] = T(B, dims[i]), T(B, 2 * L), T(2 * L), T(B, L)
        for key in ('mu', 'lv', 'z', 'eps'):
            k[f'{key}{i}'] = T(B, L)
            getattr(d, key)[i] = nv.ptr(k[f'{key}{i}'])
        d.ml[i], d.head_bias[i] = nv.ptr(k[f'ml{i}']), nv.ptr(k[f'hb{i}'])
        d.eps_in[i] = nv.ptr(k[f'eps_in{i}']) if variant == 'eps' else None
        if variant != 'nodec':
            d.g1[i], d.dec0_W[i], d.dec0_b[i], d.d[i] = nv.ptr(k[f'g1{i}']), nv.ptr(k[f'W{i}']), nv.ptr(k[f'b{i}']), dims[i]
        d.dcomb[i], d.dml[i], d.dbias_head[i] = nv.ptr(k[f'dcomb{i}']), nv.ptr(k[f'dml{i}']), nv.ptr(k[f'db{i}'])
        d.comb_alias[i] = nv.ptr(k[f'alias{i}'])
        if variant != 'nobf':
            bf = dict(device='cuda', dtype=torch.bfloat16)
            k[f'cb{i}'], k[f'cT{i}'] = torch.zeros(B, L, **bf), torch.zeros(L, B, **bf)
            k[f'db16{i}'], k[f'dT16{i}'] = torch.zeros(B, 2 * L, **bf), torch.zeros(2 * L, B, **bf)
            d.comb_bf16[i], d.combT_bf16[i] = nv.ptr(k[f'cb{i}']), nv.ptr(k[f'cT{i}'])
            d.dml_bf16[i], d.dmlT_bf16[i] = nv.ptr(k[f'db16{i}']), nv.ptr(k[f'dT16{i}'])
    d.ml_nslab, d.ml_slab_stride = nslab, B * 2 * L
    d.sigma, d.hyper, d.partials, d.comb = nv.ptr(k['sigma']), nv.ptr(k['hyper']), nv.ptr(k['partials']), nv.ptr(k['comb'])
    d.dcomb_nslab, d.dcomb_slab_stride = 2, B * L
    d.dsigma, d.rec_partials, d.n_rec_partials, d.losses = nv.ptr(k['dsigma']), nv.ptr(k['rec']), 5, nv.ptr(k['losses'])
    k['ticket'] = torch.zeros(4, dtype=torch.int32, device='cuda')
    d.colpart, d.accumulate, d.rng_stream, d.ticket = nv.ptr(k['colpart']), 0, 100, nv.ptr(k['ticket'])
    return d
state = torch.tensor([99, 5, 0, 0], dtype=torch.int64, device='cuda')
flush = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
def timeit(fn, n=30, cold=False):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if cold: flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for variant in os.environ.get('VARIANTS', 'full,nodec,eps,nobf').split(','):
    d = desc(variant)
    for cold in (False, True):
        print(f'{variant:6s} cold={int(cold)}  fwd {timeit(lambda: nv.latent_fwd(d, state), cold=cold):7.1f} us   '
              f'bwd+final {timeit(lambda: nv.latent_bwd(d), cold=cold):7.1f} us', flush=True)
