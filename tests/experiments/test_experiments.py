"""Tests of the EXPERIMENTS build (libjamie_hip_exp.so = product + -DJAMIE_EXPERIMENTS): kernels that were built and measured
slower than the product path, checked against the product kernels they would replace.  Run in a child process that loads that
library (tests/test_hip_kernels.py::test_experiments_build_passes_its_own_suite; or by hand:
JAMIE_LIB=jamie_amd/libjamie_hip_exp.so pytest tests/experiments -m gpu); skipped under the product library."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nv():
    from jamie_amd import _native
    _native.require_gpu()
    return _native


@pytest.fixture(scope='module')
def ex(nv):
    from jamie_amd import experiments
    if not experiments.available():
        pytest.skip('product library loaded: these entry points exist in the experiments build only')
    return experiments


def close(a, ref, rtol=1e-4, atol=1e-5, msg=''):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol, err_msg=msg)


def _bf16(t):
    return t.to(torch.bfloat16)


# ---- Linear forward + BatchNorm + LeakyReLU + dropout in one launch (in-launch split-K hand-off): jamie_gemm_bf16_bn ----
def _fused_bn_case(nv, ex, B, shapes, cfg, sks, p, mode, explicit_mask, rounds=3, seed=0):
    """`shapes` = [(N, K)] per modality.  The fused launch against the two launches it replaces (jamie_gemm_bf16 with the same
    tile configuration and slabs, then jamie_bn_act_fwd) on the SAME buffers, several rounds with fresh inputs (the second round
    on finds the previous round's slab lines in the caches: the hand-off must not read them): bit for bit."""
    g = torch.Generator().manual_seed(seed)
    state = torch.tensor([7, 3, 0, 0], dtype=torch.int64, device='cuda')
    n_strips = sum((N + 127) // 128 for N, _ in shapes)
    tickets = torch.zeros(4 + 2 * n_strips, dtype=torch.int32, device='cuda')
    bufs = []
    for (N, K), sk in zip(shapes, sks):
        bufs.append(dict(N=N, K=K, sk=sk,
                         a=torch.empty(B, K, dtype=torch.bfloat16, device='cuda'), W=torch.empty(N, K, dtype=torch.bfloat16, device='cuda'),
                         bias=torch.empty(N, device='cuda'), gamma=torch.empty(N, device='cuda'), beta=torch.empty(N, device='cuda'),
                         h=[torch.zeros(sk, B, N, device='cuda') for _ in range(2)],
                         out=[torch.zeros(B, N, dtype=torch.bfloat16, device='cuda') for _ in range(2)],
                         rm=[torch.zeros(N, device='cuda') for _ in range(2)], rv=[torch.ones(N, device='cuda') for _ in range(2)],
                         sm=[torch.zeros(N, device='cuda') for _ in range(2)], si=[torch.zeros(N, device='cuda') for _ in range(2)],
                         mask=torch.zeros(B, N, dtype=torch.uint8, device='cuda') if (explicit_mask and p > 0) else None))

    def problems(which):
        gp, bp = [], []
        for i, b in enumerate(bufs):
            gp.append(nv.gemm_problem(b['a'], b['W'], b['h'][which], B, b['N'], b['K'], b['K'], b['K'], b['N'], bias=b['bias'],
                                      splitk=b['sk'], slab_stride=B * b['N']))
            pr = nv.BnFwdProblem()
            pr.h, pr.nslab, pr.slab_stride = nv.ptr(b['h'][which]), b['sk'], B * b['N']
            pr.gamma, pr.beta = nv.ptr(b['gamma']), nv.ptr(b['beta'])
            pr.running_mean, pr.running_var = nv.ptr(b['rm'][which]), nv.ptr(b['rv'][which])
            pr.save_mean, pr.save_invstd = nv.ptr(b['sm'][which]), nv.ptr(b['si'][which])
            pr.out, pr.mask, pr.out_bf16 = None, nv.ptr(b['mask']), nv.ptr(b['out'][which])
            pr.B, pr.N, pr.rng_stream = B, b['N'], 10 + 8 * i
            bp.append(pr)
        return gp, bp
    for rnd in range(rounds):
        for b in bufs:
            b['a'].copy_(torch.randn(B, b['K'], generator=g).to(torch.bfloat16))
            b['W'].copy_((torch.randn(b['N'], b['K'], generator=g) * b['K'] ** -0.5).to(torch.bfloat16))
            b['bias'].copy_(torch.randn(b['N'], generator=g))
            b['gamma'].copy_(torch.rand(b['N'], generator=g) + .5)
            b['beta'].copy_(torch.randn(b['N'], generator=g) * .3)
            if b['mask'] is not None:
                b['mask'].copy_((torch.rand(B, b['N'], generator=g) >= p).to(torch.uint8))
        gp, bp = problems(0)
        nv.gemm_bf16(gp, cfg)
        nv.bn_act_fwd(bp, p, state)
        gp, bp = problems(1)
        ex.gemm_bf16_bn(gp, bp, cfg, p, state, tickets, mode)
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0, tickets.tolist()           # counters back to zero, no timeout word
        for i, b in enumerate(bufs):
            for k in ('out', 'rm', 'rv', 'sm', 'si'):
                assert torch.equal(b[k][0], b[k][1]), (rnd, i, k)
            assert torch.equal(b['h'][0][0], b['h'][1][0]), (rnd, i, 'summed pre-activation')
            if p > 0:
                kept = float((b['out'][1].float() != 0).float().mean())
                assert abs(kept - (1 - p)) < 0.02
        state[1] += 1
    # and against torch on the same bf16 operands (the two-launch path is itself tested above; this pins the pair)
    b = bufs[0]
    want = b['a'].float() @ b['W'].float().t() + b['bias']
    close(b['h'][1][0], want.cpu(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('B,shapes,cfg,sks,p,explicit', [
    (512, [(4000, 2000), (2000, 1000)], 31, (3, 2), 0.6, False),       # config 2's enc0 / dec1 launch: 256 workgroups, 6 and 4 per strip
    (512, [(2000, 4000), (1000, 2000)], 32, (3, 2), 0.6, True),        # enc1: 128 x 128 tiles, 12 and 8 per strip
    (512, [(1000, 504)], 31, (1,), 0.0, False),                        # no split: 2 workgroups per strip; ragged last strip (1000 = 7 x 128 + 104)
    (256, [(520, 264), (264, 520)], 31, (2, 1), 0.25, True),           # one M tile; strips with 8 and 1 sub-strips in use
    (200, [(264, 136)], 32, (2,), 0.6, False),                         # rows beyond the batch in the last M tile
    (512, [(10000, 5000)], 31, (2,), 0.6, False),                      # config 5's width: 79 strips, 316 workgroups (> 256 CUs: two rounds)
])
def test_gemm_bf16_fused_batchnorm_equals_two_launches(nv, ex, monkeypatch, B, shapes, cfg, sks, p, explicit, mode):
    monkeypatch.setenv('JAMIE_BN_CQ', '4')       # (the fused launch runs 16-column strips: the same summation order as these)
    _fused_bn_case(nv, ex, B, shapes, cfg, sks, p, mode, explicit, seed=B + cfg)


@pytest.mark.parametrize('shapes', [[(512, 64, 2000), (512, 64, 1000)],         # config 2's heads forward (mu | logvar), both modalities
                                    [(512, 32, 2000), (512, 32, 1000)],         # ... and its d comb product
                                    [(200, 40, 264)], [(512, 128, 5000)], [(37, 8, 104), (64, 128, 8)], [(512, 32, 2000)] * 4])
def test_gemm_bf16_skinny(nv, ex, shapes):
    """jamie_gemm_bf16_skinny (a 32 x 32 output tile per workgroup, a K slice per wave, fragments straight from global memory):
    against fp32 torch on the same bf16 operands; ragged M / N / K (K a multiple of 8 only), grouped problems, with and
    without bias; integer-valued operands are reproduced exactly."""
    g = torch.Generator().manual_seed(len(shapes) + shapes[0][2])
    probs, keep = [], []
    for j, (M, N, K) in enumerate(shapes):
        A = _bf16(torch.randn(M, K, generator=g)).cuda()
        Bm = _bf16(torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        bias = torch.randn(N, generator=g).cuda() if j % 2 == 0 else None
        Cm = torch.full((M, N), 7.0, device='cuda')
        probs.append(nv.gemm_problem(A, Bm, Cm, M, N, K, K, K, N, bias=bias))
        keep.append((A, Bm, bias, Cm))
    ex.gemm_bf16_skinny(probs)
    for A, Bm, bias, Cm in keep:
        want = A.float() @ Bm.float().t() + (bias if bias is not None else 0)
        close(Cm, want.cpu(), rtol=2e-5, atol=2e-5)
    # exact integers, asymmetric operands (a swapped row / column map or a dropped K slice shows)
    M, N, K = 96, 64, 272
    A = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
    Bm = (torch.arange(N * K).reshape(N, K) % 5 - 2).float()
    Cm = torch.zeros(M, N, device='cuda')
    ex.gemm_bf16_skinny([nv.gemm_problem(A.to(torch.bfloat16).cuda(), Bm.to(torch.bfloat16).cuda(), Cm, M, N, K, K, K, N)])
    assert torch.equal(Cm.cpu(), A @ Bm.t())




@pytest.mark.parametrize('case', ['dec2', 'dec1', 'dw_only', 'ragged', 'fp32_dw_riders'])
def test_persistent_ring_equals_one_workgroup_per_tile(nv, ex, case):
    """jamie_gemm_bf16_ring (persistent workgroups: loader waves stream a static tile list through an LDS ring, consumer waves
    multiply and store; hand-off through LDS words) against jamie_gemm_bf16 (configuration 29) on the grouped backward launch of
    a layer: dX on W as stored (split-K slabs, ragged last k-step), dW on the activations as stored (bf16 results, per-tile sums
    of squares), ragged M / N edges: bit for bit, and the hand-off error word stays zero."""
    g = torch.Generator().manual_seed(5)
    T = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).cuda()      # noqa: E731
    B = 512
    if case == 'dec2':
        dims, sks = [(2000, 4000), (1000, 2000)], (1, 1)
    elif case == 'dec1':
        dims, sks = [(4000, 2000), (2000, 1000)], (2, 1)
    elif case == 'ragged':
        B, dims, sks = 200, [(264, 520), (136, 264)], (1, 1)
    else:
        dims, sks = [(4000, 2000), (2000, 1000)], None
    n_wg = torch.cuda.get_device_properties(0).multi_processor_count

    def build(which):
        torch.manual_seed(0)
        probs, outs = [], []
        gg = torch.Generator().manual_seed(11)
        R = lambda *s: torch.randn(*s, generator=gg).to(torch.bfloat16).cuda()      # noqa: E731
        if sks is not None:
            for (nout, nin), s1 in zip(dims, sks):
                o = torch.zeros(s1, B, nin, device='cuda')
                outs.append(o)
                probs.append(nv.gemm_problem(R(B, nout), R(nout, nin), o, B, nin, nout, nout, nin, nin, splitk=s1, slab_stride=B * nin, b_tr=True))
        for (nout, nin) in dims:
            bf = case != 'fp32_dw_riders'
            o = torch.zeros(nout, nin, device='cuda', dtype=torch.bfloat16 if bf else torch.float32)
            part = torch.zeros(((nout + 127) // 128) * ((nin + 127) // 128), device='cuda')
            outs += [o, part]
            probs.append(nv.gemm_problem(R(B, nout), R(B, nin), o, nout, nin, B, nout, nin, nin, a_tr=True, b_tr=True, store_nt=True,
                                         c_bf16=bf, partial=part))
        return probs, outs
    p0, o0 = build(0)
    p1, o1 = build(1)
    err = torch.zeros(4, dtype=torch.int32, device='cuda')
    sched = ex.gemm_bf16_ring_plan(p1, n_wg)
    assert sched is not None
    ranges0 = ranges1 = None
    if case == 'fp32_dw_riders':          # + the range-norm riders (sums of squares of gradient ranges, bf16 copies, the step counter)
        gflat = torch.randn(300000, device='cuda')
        rg = nv.SqRanges([(0, 70000), (100000, 4096 * 3 + 5), (250000, 17)])
        st0 = torch.tensor([1, 5, 0, 0], dtype=torch.int64, device='cuda'); st1 = st0.clone()
        pa0 = torch.zeros(rg.blocks, device='cuda'); pa1 = torch.zeros(rg.blocks, device='cuda')
        g16a = torch.zeros(300000, device='cuda', dtype=torch.bfloat16); g16b = torch.zeros_like(g16a)
        ranges0 = (gflat, g16a, rg, pa0, st0, None)
        ranges1 = (gflat, g16b, rg, pa1, st1, None)
    nv.gemm_bf16(p0, 29, ranges0)
    for rep in range(3):                   # (repeated: hot and cold operands, generations of the hand-off words)
        ex.gemm_bf16_ring(p1, sched, n_wg, err, ranges1 if rep == 0 else None)
    torch.cuda.synchronize()
    assert int(err[0].item()) == 0
    for a, b in zip(o0, o1):
        if a.dim() == 1:      # (per-tile sums of squares: the product kernel's lean epilogue adds the same squares in another association)
            assert torch.allclose(a, b, rtol=1e-6), (case, float((a - b).abs().max()))
        else:
            assert torch.equal(a, b), (case, float((a.float() - b.float()).abs().max()))
    if ranges0 is not None:
        # (the riders run on 768 instead of 512 threads here: the same sums of squares in another order)
        assert torch.allclose(pa0, pa1, rtol=1e-5) and torch.equal(g16a, g16b) and int(st1[1]) == 6 == int(st0[1])
    # against fp32 torch on the same operands (pins the pair)
    A, Bm = p1[-1]._keep[0], p1[-1]._keep[1]
    close(o1[-2].float(), (A.float().t() @ Bm.float()).cpu(), rtol=2e-2, atol=2e-2)
