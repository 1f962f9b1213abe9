"""Per-kernel parity tests of libjamie_hip.so through the C ABI (ctypes), against fp64/fp32 PyTorch CPU
references and the oracle.  Run on the MI355X box:  pytest -m gpu"""
import numpy as np
import pytest
import torch

from oracle import jamie_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nv():
    from jamie_amd import _native
    _native.require_gpu()
    return _native


def dev(t):
    return t.to('cuda').contiguous()


def close(a, ref, rtol=1e-4, atol=1e-5, msg=''):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol, err_msg=msg)


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(64, 64, 32), (512, 256, 128), (100, 70, 50), (33, 130, 17), (512, 64, 1000), (17, 3, 5),
               (130, 257, 264), (256, 1000, 512)]


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
def test_gemm_layouts(nv, layout, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = a.double() @ b.double() + (bias.double() if layout == 'NT' else 0)
    out = torch.full((M, N), float('nan'), device='cuda')
    if layout == 'NT':
        A, Bm = dev(a), dev(b.t())
        pr = nv.gemm_problem(A, Bm, out, M, N, K, K, K, N, bias=dev(bias))
        nv.gemm([pr], nv.NT)
    elif layout == 'NN':
        A, Bm = dev(a), dev(b)
        nv.gemm([nv.gemm_problem(A, Bm, out, M, N, K, K, N, N)], nv.NN)
    else:
        A, Bm = dev(a.t()), dev(b)
        nv.gemm([nv.gemm_problem(A, Bm, out, M, N, K, M, N, N)], nv.TN)
    torch.cuda.synchronize()
    scale = float(np.sqrt(K))
    close(out, ref, rtol=1e-5, atol=2e-6 * scale)


@pytest.mark.parametrize('cfg', [4, 10, 11, 12, 15, 16, 17, 18, 20, 21])
@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
def test_gemm_f32_large_tile_configurations(nv, layout, cfg):
    """The register-staged fp32 kernel on its larger tiles (128x128 on 8 / 16 waves, 64x128, 128x64, 256x128, 128x256) in every
    operand layout, ragged edges and split-K slabs included, against fp64.  17 / 18 = 12 / 1 with the k-loop whose barrier sits in
    the middle of a k-step (one, two, odd and even numbers of k-steps): the same products in the same order, bit for bit."""
    twin = {17: 12, 18: 1}.get(cfg)
    for (M, N, K, sk) in ((512, 520, 264, 1), (300, 200, 100, 1), (512, 384, 1000, 3), (130, 72, 20, 1), (64, 200, 40, 1), (100, 100, 65, 1)):
        g = torch.Generator().manual_seed(M + N + K + cfg)
        a, b = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
        out = torch.full((sk, M, N), float('nan'), device='cuda')
        kw = dict(splitk=sk, slab_stride=M * N) if sk > 1 else {}
        if layout == 'NT':
            nv.gemm([nv.gemm_problem(dev(a), dev(b.t()), out, M, N, K, K, K, N, **kw)], nv.NT, cfg)
        elif layout == 'NN':
            nv.gemm([nv.gemm_problem(dev(a), dev(b), out, M, N, K, K, N, N, **kw)], nv.NN, cfg)
        else:
            nv.gemm([nv.gemm_problem(dev(a.t()), dev(b), out, M, N, K, M, N, N, **kw)], nv.TN, cfg)
        torch.cuda.synchronize()
        close(out.sum(0), a.double() @ b.double(), rtol=1e-5, atol=2e-6 * float(np.sqrt(K)), msg=f'{layout} cfg {cfg} {(M, N, K, sk)}')
        if twin is not None:
            ref = torch.full((sk, M, N), float('nan'), device='cuda')
            if layout == 'NT':
                nv.gemm([nv.gemm_problem(dev(a), dev(b.t()), ref, M, N, K, K, K, N, **kw)], nv.NT, twin)
            elif layout == 'NN':
                nv.gemm([nv.gemm_problem(dev(a), dev(b), ref, M, N, K, K, N, N, **kw)], nv.NN, twin)
            else:
                nv.gemm([nv.gemm_problem(dev(a.t()), dev(b), ref, M, N, K, M, N, N, **kw)], nv.TN, twin)
            assert torch.equal(out, ref), f'{layout} cfg {cfg} != cfg {twin} at {(M, N, K, sk)}'


@pytest.mark.parametrize('cfg', [1, 12, 17, 20, 21])
def test_gemm_f32_lean_epilogue_is_exact_on_integers(nv, cfg):
    """The fp32 kernel's lean store path (4-byte buffer stores with a scalar-register row offset, the value's register reused two
    instructions later): integer-valued operands, exact products, every element, interior and edge tiles, slabs and a plain
    result with per-tile sums of squares; five launches."""
    for (M, N, K, sk) in ((512, 384, 96, 1), (300, 200, 64, 1), (512, 520, 256, 2)):
        g = torch.Generator().manual_seed(M + N + K + cfg)
        a = torch.randint(-4, 5, (M, K), generator=g).float()
        w = torch.randint(-4, 5, (N, K), generator=g).float()
        ref = a @ w.t()
        A, W = dev(a), dev(w)
        for rep in range(5):
            out = torch.full((sk, M, N), float('nan'), device='cuda')
            kw = dict(splitk=sk, slab_stride=M * N) if sk > 1 else {}
            nv.gemm([nv.gemm_problem(A, W, out, M, N, K, K, K, N, **kw)], nv.NT, cfg)
            assert torch.equal(out.sum(0).cpu(), ref), (cfg, M, N, K, sk, rep)


@pytest.mark.parametrize('x3', [20, 21])
@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
def test_gemm_f32_three_bf16_pieces_keep_fp32(nv, layout, x3):
    """Configurations 20 / 21 (128 x 128 / 256 x 128 tiles; the fp32 products on the bf16 matrix pipe, every element cut into three bf16 pieces): (a) the cut is
    EXACT -- a product with a 0/1 selection matrix returns every 24-bit significand bit of the other operand, in either operand
    position and every layout; (b) on config 2's layer shapes its distance from fp64 stays within a small factor of the fp32
    pipe's own (configuration 17), split-K slabs included."""
    g = torch.Generator().manual_seed(20 + len(layout))
    M, N, K = 256, 384, 384

    def run(a, b, cfg, sk=1):                 # a [M, K], b [K, N] on the host
        m, k = a.shape; n = b.shape[1]
        out = torch.full((sk, m, n), float('nan'), device='cuda')
        kw = dict(splitk=sk, slab_stride=m * n) if sk > 1 else {}
        if layout == 'NT':
            nv.gemm([nv.gemm_problem(dev(a), dev(b.t()), out, m, n, k, k, k, n, **kw)], nv.NT, cfg)
        elif layout == 'NN':
            nv.gemm([nv.gemm_problem(dev(a), dev(b), out, m, n, k, k, n, n, **kw)], nv.NN, cfg)
        else:
            nv.gemm([nv.gemm_problem(dev(a.t()), dev(b), out, m, n, k, m, n, n, **kw)], nv.TN, cfg)
        torch.cuda.synchronize()
        return out.sum(0).cpu()
    a = torch.randn(M, K, generator=g) * torch.exp(4 * torch.randn(M, K, generator=g))          # wide range of exponents
    perm = torch.randperm(K, generator=g)
    sel = torch.zeros(K, N); sel[perm, torch.arange(K)] = 1.0                                       # out[:, j] = a[:, perm[j]]
    assert torch.equal(run(a, sel, x3)[:, :K], a[:, perm]), 'A operand: a piece of the cut is lost'
    b = torch.randn(K, N, generator=g) * torch.exp(4 * torch.randn(K, N, generator=g))
    selm = torch.zeros(M, K); selm[torch.arange(M), perm[:M]] = 1.0
    assert torch.equal(run(selm, b, x3), b[perm[:M]]), 'B operand: a piece of the cut is lost'
    for (m, n, k, sk) in ((512, 4000, 2000, 3), (512, 1000, 2000, 1)):
        a, b = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g) / np.sqrt(k)
        ref = a.double() @ b.double()
        e3 = (run(a, b, x3, sk).double() - ref).abs().max().item()
        e1 = (run(a, b, 17, sk).double() - ref).abs().max().item()
        assert e3 <= 4 * e1 + 1e-7, (layout, m, n, k, sk, e3, e1)


@pytest.mark.parametrize('cfg', [1, 12, 17, 18, 20, 21])
@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
def test_gemm_f32_rows_that_end_inside_a_float4(nv, layout, cfg):
    """Buffer-descriptor path with in-row tails (the `FAST = 1` instantiation: 16-byte aligned bases and leading dimensions that
    are multiples of 4, but K / M / N are not): operands live in padded storage whose padding holds NaN -- a value read from
    beyond a row's end poisons the result -- against fp64."""
    M, N, K = 130, 70, 66
    g = torch.Generator().manual_seed(11 * cfg + len(layout))
    a, b = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)

    def padded(t, ld):            # [rows, ld] storage, NaN beyond the logical columns
        buf = torch.full((t.shape[0], ld), float('nan'))
        buf[:, :t.shape[1]] = t
        return dev(buf)
    out = torch.full((M, N), float('nan'), device='cuda')
    if layout == 'NT':
        A, Bm = padded(a, 68), padded(b.t().contiguous(), 68)
        nv.gemm([nv.gemm_problem(A, Bm, out, M, N, K, 68, 68, N)], nv.NT, cfg)
    elif layout == 'NN':
        A, Bm = padded(a, 68), padded(b, 72)
        nv.gemm([nv.gemm_problem(A, Bm, out, M, N, K, 68, 72, N)], nv.NN, cfg)
    else:
        A, Bm = padded(a.t().contiguous(), 132), padded(b, 72)
        nv.gemm([nv.gemm_problem(A, Bm, out, M, N, K, 132, 72, N)], nv.TN, cfg)
    torch.cuda.synchronize()
    close(out, a.double() @ b.double(), rtol=1e-5, atol=2e-6 * float(np.sqrt(K)), msg=f'{layout} cfg {cfg}')


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES + [(512, 2000, 1000), (130, 72, 1002), (64, 64, 20)])
@pytest.mark.parametrize('cfg', [7, 8, 9])
def test_gemm_f32_lds_dma_nt(nv, M, N, K, cfg):
    """The LDS-DMA NT kernel (64x64x32, 3 / 2 / 4 buffers): exact-fp32 products against fp64, bit-exact on small integers,
    ragged M / N, partial k-tiles, K % 4 != 0 falls back to the register-staged kernel, split-K slabs, MSE and
    eval-BatchNorm epilogues, grouped problems."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + cfg)
    a, w, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    out = torch.full((M, N), float('nan'), device='cuda')
    nv.gemm([nv.gemm_problem(dev(a), dev(w), out, M, N, K, K, K, N, bias=dev(bias))], nv.NT, cfg)
    close(out, a.double() @ w.double().t() + bias.double(), rtol=1e-5, atol=4e-6 * float(np.sqrt(K)))   # 1M outputs: 4.5 sigma
    ai = torch.randint(-4, 5, (M, K), generator=g).float()
    wi = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 13) - 6
    out2 = torch.ones(M, N, device='cuda')
    nv.gemm([nv.gemm_problem(dev(ai), dev(wi), out2, M, N, K, K, K, N, accumulate=True)], nv.NT, cfg)
    assert torch.equal(out2.cpu(), ai @ wi.t() + 1.0)
    if K >= 128:
        slabs = torch.full((3, M, N), float('nan'), device='cuda')
        x1, w1 = torch.randn(40, 66 * 4, generator=g), torch.randn(72, 66 * 4, generator=g)
        o1 = torch.zeros(40, 72, device='cuda')
        nv.gemm([nv.gemm_problem(dev(a), dev(w), slabs, M, N, K, K, K, N, bias=dev(bias), splitk=3, slab_stride=M * N),
                 nv.gemm_problem(dev(x1), dev(w1), o1, 40, 72, 264, 264, 264, 72)], nv.NT, cfg)
        close(slabs.sum(0), a.double() @ w.double().t() + bias.double(), rtol=1e-5, atol=1e-4)
        close(o1, x1.double() @ w1.double().t(), rtol=1e-5, atol=5e-5)


def test_gemm_f32_lds_dma_epilogues(nv):
    import math
    B, d, K = 100, 150, 80
    g = torch.Generator().manual_seed(9)
    e2, W, b, X = (torch.randn(B, K, generator=g), torch.randn(d, K, generator=g), torch.randn(d, generator=g),
                   torch.randn(B, d, generator=g))
    part = torch.zeros(math.ceil(B / 64) * math.ceil(d / 64), device='cuda')
    out = torch.zeros(B, d, device='cuda')
    scale = 2.0 / (B * d)
    nv.gemm([nv.gemm_problem(dev(e2), dev(W), out, B, d, K, K, K, d, bias=dev(b), epi=nv.EPI_MSE,
                             aux=(dev(X), None, None, None), aux_ld=d, partial=part, scale=scale, pscale=1.0 / (B * d))], nv.NT, 7)
    diff = e2.double() @ W.double().t() + b.double() - X.double()
    close(out, diff * scale, rtol=1e-5, atol=1e-7)
    close(part.sum(), (diff ** 2).mean(), rtol=1e-5, atol=0)
    rm, rv = torch.randn(d, generator=g), torch.rand(d, generator=g) + .5
    ga, be = torch.randn(d, generator=g), torch.randn(d, generator=g)
    nv.gemm([nv.gemm_problem(dev(e2), dev(W), out, B, d, K, K, K, d, bias=dev(b), epi=nv.EPI_BN_EVAL,
                             aux=(dev(rm), dev(rv), dev(ga), dev(be)), slope=0.01, eps=1e-5)], nv.NT, 7)
    h = torch.nn.functional.linear(e2.double(), W.double(), b.double())
    ref = torch.nn.functional.leaky_relu(
        torch.nn.functional.batch_norm(h, rm.double(), rv.double(), ga.double(), be.double(), False, 0.1, 1e-5), 0.01)
    close(out, ref, rtol=1e-4, atol=1e-5)


def test_gemm_identity_asymmetric(nv):
    """A = I with an asymmetric B catches a swapped C/D register map (cdna_hip_programming.md §3)."""
    n = 96
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    out = torch.zeros(n, n, device='cuda')
    nv.gemm([nv.gemm_problem(dev(torch.eye(n)), dev(B), out, n, n, n, n, n, n)], nv.NN)
    assert torch.equal(out.cpu(), B)
    nv.gemm([nv.gemm_problem(dev(torch.eye(n)), dev(B.t()), out, n, n, n, n, n, n)], nv.NT)
    assert torch.equal(out.cpu(), B)
    nv.gemm([nv.gemm_problem(dev(torch.eye(n)), dev(B), out, n, n, n, n, n, n)], nv.TN)
    assert torch.equal(out.cpu(), B)


@pytest.mark.parametrize('splitk', [2, 3, 7])
def test_gemm_splitk_slabs(nv, splitk):
    M, N, K = 200, 48, 1000
    g = torch.Generator().manual_seed(splitk)
    a, w, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    slabs = torch.full((splitk, M, N), float('nan'), device='cuda')
    nv.gemm([nv.gemm_problem(dev(a), dev(w), slabs, M, N, K, K, K, N, bias=dev(bias), splitk=splitk,
                             slab_stride=M * N)], nv.NT)
    close(slabs.sum(0), a.double() @ w.double().t() + bias.double(), rtol=1e-5, atol=1e-4)


def test_gemm_grouped_and_row_gather(nv):
    g = torch.Generator().manual_seed(5)
    table = torch.randn(300, 72, generator=g)
    rows = torch.randint(0, 300, (64,), generator=g, dtype=torch.int32)
    w0, w1 = torch.randn(144, 72, generator=g), torch.randn(40, 66, generator=g)
    x1 = torch.randn(64, 66, generator=g)
    o0 = torch.zeros(64, 144, device='cuda')
    o1 = torch.zeros(64, 40, device='cuda')
    nv.gemm([nv.gemm_problem(dev(table), dev(w0), o0, 64, 144, 72, 72, 72, 144, a_rows=dev(rows)),
             nv.gemm_problem(dev(x1), dev(w1), o1, 64, 40, 66, 66, 66, 40)], nv.NT)
    close(o0, table[rows.long()].double() @ w0.double().t(), rtol=1e-5, atol=2e-5)
    close(o1, x1.double() @ w1.double().t(), rtol=1e-5, atol=2e-5)


def test_gemm_mse_epilogue(nv):
    B, d, K = 100, 150, 80
    g = torch.Generator().manual_seed(9)
    e2, W, b, X = (torch.randn(B, K, generator=g), torch.randn(d, K, generator=g), torch.randn(d, generator=g),
                   torch.randn(B, d, generator=g))
    import math
    bm, bn = nv.gemm_tile(nv.NT, B, d, K)
    tiles = math.ceil(B / bm) * math.ceil(d / bn)
    part = torch.zeros(tiles, device='cuda')
    out = torch.zeros(B, d, device='cuda')
    scale = 2.0 / (B * d)
    nv.gemm([nv.gemm_problem(dev(e2), dev(W), out, B, d, K, K, K, d, bias=dev(b), epi=nv.EPI_MSE,
                             aux=(dev(X), None, None, None), aux_ld=d, partial=part, scale=scale,
                             pscale=1.0 / (B * d))], nv.NT)
    diff = e2.double() @ W.double().t() + b.double() - X.double()
    close(out, diff * scale, rtol=1e-5, atol=1e-7)
    close(part.sum(), (diff ** 2).mean(), rtol=1e-5, atol=0)


def test_gemm_bn_eval_epilogue(nv):
    n, K, N = 77, 40, 90
    g = torch.Generator().manual_seed(11)
    x, W, b = torch.randn(n, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    rm, rv = torch.randn(N, generator=g), torch.rand(N, generator=g) + .5
    ga, be = torch.randn(N, generator=g), torch.randn(N, generator=g)
    out = torch.zeros(n, N, device='cuda')
    nv.gemm([nv.gemm_problem(dev(x), dev(W), out, n, N, K, K, K, N, bias=dev(b), epi=nv.EPI_BN_EVAL,
                             aux=(dev(rm), dev(rv), dev(ga), dev(be)), slope=0.01, eps=1e-5)], nv.NT)
    h = torch.nn.functional.linear(x.double(), W.double(), b.double())
    ref = torch.nn.functional.leaky_relu(
        torch.nn.functional.batch_norm(h, rm.double(), rv.double(), ga.double(), be.double(), False, 0.1, 1e-5), 0.01)
    close(out, ref, rtol=1e-4, atol=1e-5)


def test_gemm_accumulate(nv):
    M, N, K = 70, 90, 64
    g = torch.Generator().manual_seed(2)
    a, b = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    out = dev(c0.clone())
    nv.gemm([nv.gemm_problem(dev(a), dev(b), out, M, N, K, M, N, N, accumulate=True)], nv.TN)
    close(out, c0.double() + a.double().t() @ b.double(), rtol=1e-5, atol=2e-5)


def test_gemm_bad_arguments_fail_loudly(nv):
    a = torch.zeros(4, 4, device='cuda')
    with pytest.raises(nv.JamieHipError):
        nv.gemm([nv.gemm_problem(a, a, a, 4, 4, 4, 2, 4, 4)], nv.NT)           # lda < K
    with pytest.raises(nv.JamieHipError):
        nv.gemm([nv.gemm_problem(a, a, a, 4, 4, 4, 4, 4, 4, epi=nv.EPI_MSE)], nv.NT)   # aux missing
    with pytest.raises(nv.JamieHipError):
        nv.gemm_problem(torch.zeros(4, 4), a, a, 4, 4, 4, 4, 4, 4)               # CPU tensor


# ------------------------------------------------------------------------------------------------
# BatchNorm + LeakyReLU + Dropout
# ------------------------------------------------------------------------------------------------
def _bn_ref(h, gamma, beta, mask, p, da):
    h = h.double().requires_grad_(True)
    gamma = gamma.double().requires_grad_(True)
    beta = beta.double().requires_grad_(True)
    rm, rv = torch.zeros(h.shape[1], dtype=torch.float64), torch.ones(h.shape[1], dtype=torch.float64)
    y = torch.nn.functional.batch_norm(h, rm, rv, gamma, beta, True, 0.1, 1e-5)
    y = torch.nn.functional.leaky_relu(y, 0.01)
    if p > 0:
        y = y * (mask.double() / (1 - p))
    y.backward(da.double())
    return y.detach(), rm, rv, h.grad, gamma.grad, beta.grad


@pytest.mark.parametrize('B,N,p,nslab', [(32, 40, 0.6, 1), (512, 100, 0.6, 2), (100, 33, 0.0, 1), (700, 24, 0.25, 3),
                                          (1024, 100, 0.6, 2), (1000, 36, 0.0, 1), (1032, 24, 0.25, 1)])
def test_bn_act_fwd_bwd(nv, B, N, p, nslab):
    g = torch.Generator().manual_seed(B + N)
    hs = torch.randn(nslab, B, N, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(N, generator=g) + .5, torch.randn(N, generator=g)
    mask = (torch.rand(B, N, generator=g) >= p).to(torch.uint8)
    da_s = torch.randn(nslab, B, N, generator=g)
    h_sum, da = hs.sum(0), da_s.sum(0)
    y, rm, rv, dh, dg, db = _bn_ref(h_sum, gamma, beta, mask, p, da)
    hd = dev(hs)
    out = torch.zeros(B, N, device='cuda')
    rmean, rvar = torch.zeros(N, device='cuda'), torch.ones(N, device='cuda')
    smean, sinv = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    mk = dev(mask) if p > 0 else None
    pr = nv.BnFwdProblem()
    pr.h, pr.nslab, pr.slab_stride = nv.ptr(hd), nslab, B * N
    pr.gamma, pr.beta = nv.ptr(dev(gamma)), nv.ptr(dev(beta))
    gd, bd = dev(gamma), dev(beta)
    pr.gamma, pr.beta = nv.ptr(gd), nv.ptr(bd)
    pr.running_mean, pr.running_var, pr.save_mean, pr.save_invstd = (nv.ptr(rmean), nv.ptr(rvar), nv.ptr(smean),
                                                                      nv.ptr(sinv))
    pr.out, pr.mask, pr.B, pr.N, pr.rng_stream = nv.ptr(out), nv.ptr(mk), B, N, 0
    nv.bn_act_fwd([pr], p, None)
    close(out, y, rtol=1e-4, atol=1e-5)
    close(rmean, rm, rtol=1e-5, atol=1e-6)
    close(rvar, rv, rtol=1e-5, atol=1e-6)
    close(hd[0], h_sum, rtol=1e-6, atol=1e-6)
    dad = dev(da_s)
    dgam, dbet, dbl = (torch.zeros(N, device='cuda') for _ in range(3))
    pb = nv.BnBwdProblem()
    pb.da, pb.nslab, pb.slab_stride = nv.ptr(dad), nslab, B * N
    pb.h, pb.gamma, pb.beta, pb.save_mean, pb.save_invstd = nv.ptr(hd), nv.ptr(gd), nv.ptr(bd), nv.ptr(smean), nv.ptr(sinv)
    pb.dgamma, pb.dbeta, pb.dbias_lin, pb.mask = nv.ptr(dgam), nv.ptr(dbet), nv.ptr(dbl), nv.ptr(mk)
    pb.B, pb.N, pb.rng_stream, pb.accumulate = B, N, 0, 0
    nv.bn_act_bwd([pb], p, None)
    close(dad[0], dh, rtol=1e-3, atol=2e-5)
    close(dgam, dg, rtol=1e-4, atol=1e-4)
    close(dbet, db, rtol=1e-4, atol=1e-4)
    assert dbl.abs().max().item() < 1e-3      # colsum(dh) is mathematically zero


def test_dropout_rng_consistency_and_rate(nv):
    """Philox masks: forward and backward regenerate the same mask; keep rate = 1 - p; a new step or a
    different stream gives a different mask."""
    B, N, p = 512, 256, 0.6
    h = torch.randn(1, B, N)
    hd = dev(h)
    state = torch.tensor([1234, 0, 0, 0], dtype=torch.int64, device='cuda')
    ones, zeros = torch.ones(N, device='cuda'), torch.zeros(N, device='cuda')

    def fwd(stream):
        out = torch.zeros(B, N, device='cuda')
        pr = nv.BnFwdProblem()
        pr.h, pr.nslab, pr.slab_stride, pr.gamma, pr.beta = nv.ptr(hd), 1, B * N, nv.ptr(ones), nv.ptr(zeros)
        rm, rv, sm, si = (torch.zeros(N, device='cuda') for _ in range(4))
        pr.running_mean, pr.running_var, pr.save_mean, pr.save_invstd = nv.ptr(rm), nv.ptr(rv), nv.ptr(sm), nv.ptr(si)
        pr.out, pr.mask, pr.B, pr.N, pr.rng_stream = nv.ptr(out), None, B, N, stream
        nv.bn_act_fwd([pr], p, state)
        return out, sm, si
    o1, sm, si = fwd(3)
    o2, _, _ = fwd(3)
    o3, _, _ = fwd(4)
    assert torch.equal(o1, o2)
    keep = (o1 != 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01
    assert (o1 != 0).ne(o3 != 0).float().mean().item() > 0.3
    # backward with da = 1: dy != 0 exactly where the forward kept the element
    da = torch.ones(1, B, N, device='cuda')
    dg, db = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    pb = nv.BnBwdProblem()
    pb.da, pb.nslab, pb.slab_stride = nv.ptr(da), 1, B * N
    pb.h, pb.gamma, pb.beta, pb.save_mean, pb.save_invstd = nv.ptr(hd), nv.ptr(ones), nv.ptr(zeros), nv.ptr(sm), nv.ptr(si)
    pb.dgamma, pb.dbeta, pb.dbias_lin, pb.mask = nv.ptr(dg), nv.ptr(db), None, None
    pb.B, pb.N, pb.rng_stream, pb.accumulate = B, N, 3, 0
    nv.bn_act_bwd([pb], p, state)
    # dbeta = sum_b dy = sum over kept of (1/(1-p)) * lrelu'(y)
    y = (h[0] - h[0].mean(0)) / torch.sqrt(h[0].var(0, unbiased=False) + 1e-5)
    kept = (o1 != 0).cpu()
    ref_db = (kept.float() / (1 - p) * torch.where(y > 0, 1.0, 0.01)).sum(0)
    close(db, ref_db, rtol=1e-4, atol=1e-3)
    state[1] = 1
    o4, _, _ = fwd(3)
    assert (o1 != 0).ne(o4 != 0).float().mean().item() > 0.3


# ------------------------------------------------------------------------------------------------
# latent block
# ------------------------------------------------------------------------------------------------
def _latent_case(nv, B, L, general, useF, cosine, nslab, seed):
    g = torch.Generator().manual_seed(seed)
    ml = [torch.randn(nslab, B, 2 * L, generator=g) * .5 for _ in range(2)]
    hb = [torch.randn(2 * L, generator=g) * .1 for _ in range(2)]
    eps = [torch.randn(B, L, generator=g) for _ in range(2)]
    sigma = torch.rand(2, generator=g) + .2
    corr = Fb = None
    if general:
        idx = torch.randint(0, B // 2 + 1, (B,), generator=g)
        corr = orc.p_block(None, idx.numpy(), idx.numpy())
        if useF:
            Fb = orc.row_normalise(torch.rand(B, B, generator=g) * (torch.rand(B, B, generator=g) < .3))
            corr = .5 * corr + .5 * Fb
    dcomb = [torch.randn(nslab, B, L, generator=g) * .01 for _ in range(2)]
    kl_scale, w_rec, w_al, w_f = 0.02, 1.0, 32.0 * 1.5, 0.7
    # ---- reference (fp64 autograd) ----
    mlr = [(m.sum(0) + b).double().requires_grad_(True) for m, b in zip(ml, hb)]
    sig = sigma.double().requires_grad_(True)
    mus = [m[:, :L] for m in mlr]
    lvs = [m[:, L:] for m in mlr]
    zs = [mus[i] + eps[i].double() * (torch.exp(lvs[i] / 2) + 1e-7) for i in range(2)]
    C = torch.eye(B, dtype=torch.float64) if corr is None else corr.double()
    Fm = torch.zeros(B, B, dtype=torch.float64) if Fb is None else Fb.double()
    comb = orc.combine({'sigma': sig}, zs, C)
    logv = lvs[1]
    kl = sum(-.5 * torch.mean(1 + logv[i] - mus[i].square() - logv[i].exp(), axis=1).mean(axis=0) for i in range(2))
    if cosine:
        d = [1 - torch.nn.functional.cosine_similarity(zs[i], comb[i], dim=1, eps=0) for i in range(2)]
        al = sum((d[i] ** 2).mean() / L for i in range(2))
    else:
        al = sum(((zs[i] - comb[i]) ** 2).sum(1).mean() / L for i in range(2))
    lf = torch.square(comb[0] - Fm @ comb[1]).mean()
    ext = sum((comb[i] * dcomb[i].sum(0).double()).sum() for i in range(2))    # stands for the decoder path
    total = kl_scale * kl + w_al * al + w_f * lf + ext
    total.backward()
    # ---- HIP ----
    f32 = dict(device='cuda', dtype=torch.float32)
    d = nv.Latent()
    d.B, d.L = B, L
    keep = {}
    hyper = torch.zeros(16)
    hyper[0], hyper[1], hyper[2], hyper[3] = kl_scale, w_rec, w_al, w_f
    keep['hyper'] = dev(hyper)
    for i in range(2):
        keep[f'ml{i}'], keep[f'hb{i}'], keep[f'epsin{i}'], keep[f'dc{i}'] = dev(ml[i]), dev(hb[i]), dev(eps[i]), dev(dcomb[i])
        d.ml[i], d.head_bias[i], d.eps_in[i], d.dcomb[i] = (nv.ptr(keep[f'ml{i}']), nv.ptr(keep[f'hb{i}']),
                                                            nv.ptr(keep[f'epsin{i}']), nv.ptr(keep[f'dc{i}']))
        for k in ('mu', 'lv', 'z', 'eps', 'comb', 'cz', 'H', 'ch'):
            keep[f'{k}{i}'] = torch.zeros(B, L, **f32)
            getattr(d, k)[i] = nv.ptr(keep[f'{k}{i}'])
        keep[f'dml{i}'] = torch.zeros(B, 2 * L, **f32)
        d.dml[i] = nv.ptr(keep[f'dml{i}'])
    d.ml_nslab, d.ml_slab_stride, d.dcomb_nslab, d.dcomb_slab_stride = nslab, B * 2 * L, nslab, B * L
    keep['sigma'] = dev(sigma)
    keep['corr'] = dev(corr) if corr is not None else None
    keep['F'] = dev(Fb) if Fb is not None else None
    for k in ('rsum', 'qsum'):
        keep[k] = torch.zeros(B, **f32)
    for k in ('fc1', 'fte'):
        keep[k] = torch.zeros(B, L, **f32)
    keep['partials'] = torch.zeros(16 * nv.load().jamie_max_partials(), **f32)
    keep['dsigma'] = torch.zeros(2, **f32)
    keep['losses'] = torch.zeros(8, **f32)
    keep['losses'][5] = float('inf')
    keep['rec'] = dev(torch.tensor([0.25, 0.5]))
    d.sigma, d.corr, d.Fblk, d.hyper = nv.ptr(keep['sigma']), nv.ptr(keep['corr']), nv.ptr(keep['F']), nv.ptr(keep['hyper'])
    d.rsum, d.qsum, d.fc1, d.fte, d.partials = (nv.ptr(keep['rsum']), nv.ptr(keep['qsum']), nv.ptr(keep['fc1']),
                                               nv.ptr(keep['fte']), nv.ptr(keep['partials']))
    d.dsigma, d.rec_partials, d.n_rec_partials, d.losses = nv.ptr(keep['dsigma']), nv.ptr(keep['rec']), 2, nv.ptr(keep['losses'])
    d.cosine, d.rng_stream = int(cosine), 100
    nv.latent_fwd(d, None)
    nv.latent_bwd(d)
    torch.cuda.synchronize()
    for i in range(2):
        close(keep[f'mu{i}'], mus[i], 1e-5, 1e-6, 'mu')
        close(keep[f'z{i}'], zs[i], 1e-5, 1e-6, 'z')
        close(keep[f'comb{i}'], comb[i], 1e-5, 1e-6, 'comb')
        close(keep[f'dml{i}'], mlr[i].grad, 2e-4, 2e-6, f'dml{i}')
    close(keep['dsigma'], sig.grad, 2e-4, 1e-5, 'dsigma')
    ls = keep['losses'].cpu().double()
    close(ls[0], kl_scale * kl, 1e-4, 1e-7, 'KL')
    close(ls[1], w_rec * 0.75, 1e-6, 0, 'Rec')
    close(ls[2], w_al * al, 1e-4, 1e-6, 'align')
    close(ls[3], w_f * lf, 1e-4, 1e-7, 'F')
    close(ls[4], ls[:4].sum(), 1e-6, 0)
    close(ls[5], ls[4], 1e-7, 0)


@pytest.mark.parametrize('B,L,general,useF,cosine,nslab', [
    (16, 4, False, False, False, 1), (64, 8, True, False, False, 2), (48, 4, True, True, False, 1),
    (20, 3, True, False, True, 1), (512, 32, False, False, False, 7), (300, 16, True, True, True, 2)])
def test_latent_fwd_bwd(nv, B, L, general, useF, cosine, nslab):
    _latent_case(nv, B, L, general, useF, cosine, nslab, B * 3 + L)


@pytest.mark.parametrize('M,B,L,dims,nslab,bf,acc', [
    (2, 512, 32, (2000, 1000), 3, True, False), (2, 50, 8, (72, 45), 1, False, False), (3, 130, 64, (300, 200, 136), 2, True, True),
    (4, 64, 16, (40, 33, 24, 16), 1, False, False), (2, 96, 128, (70, 52), 2, False, False),
    (2, 70, 128, (72, 52), 3, True, False), (4, 64, 16, (40, 32, 24, 16), 9, True, False), (3, 40, 64, (260, 136, 72), 8, False, True)])
def test_latent_m_fused_kernels(nv, M, B, L, dims, nslab, bf, acc):
    """The fused latent kernels of the identity-correspondence step (jamie_latent_m_fwd / _bwd; reference model.py:225-259
    with corr = I, model.py:190 decoder layer 0, jamie.py:618-668 losses) against a float64 autograd restatement:
    forward = mu / logvar / z / comb from the heads' split-K slabs, the decoder's first pre-activation g1 = comb W^T + b for
    every modality (ragged column chunks, ragged last row block) and the losses; backward = d(mu | logvar), d sigma and
    the head-bias gradients (column sums; accumulated when asked), with the bf16 copies the bf16 GEMMs read, and (feature counts
    that are multiples of 4) the heads' input gradient d a2 = d(mu | logvar) W_head (model.py:141-143 backward)."""
    f32 = dict(device='cuda', dtype=torch.float32)
    g = torch.Generator().manual_seed(M * 1000 + B + L)
    ml = [torch.randn(nslab, B, 2 * L, generator=g) * .5 for _ in range(M)]
    hb = [torch.randn(2 * L, generator=g) * .1 for _ in range(M)]
    eps = [torch.randn(B, L, generator=g) for _ in range(M)]
    sigma = torch.rand(M, generator=g) + .2
    W = [torch.randn(d, L, generator=g) / L ** .5 for d in dims]
    bd = [torch.randn(d, generator=g) * .1 for d in dims]
    Wh = [torch.randn(2 * L, d, generator=g) / d ** .5 for d in dims]      # head weights [2L, d]
    with_da2 = all(d % 4 == 0 for d in dims)
    ndc = 2
    dcomb = [torch.randn(ndc, B, L, generator=g) * 1e-3 for _ in range(M)]
    hyper = torch.zeros(16)
    hyper[0:4] = torch.tensor([0.016, 1.5, 32.0, 0.7])
    recp = torch.rand(5, generator=g)
    prev_db = [torch.randn(2 * L, generator=g) for _ in range(M)]
    # ---- float64 reference ----
    mlp = [(m.double().sum(0) + h.double()).requires_grad_(True) for m, h in zip(ml, hb)]
    sg = sigma.double().requires_grad_(True)
    mus, lvs, zs = [], [], []
    for i in range(M):
        mu, lv = mlp[i][:, :L], mlp[i][:, L:]
        mus.append(mu); lvs.append(lv)
        zs.append(mu + eps[i].double() * (torch.exp(lv / 2) + 1e-7))
    comb = sum(sg[i] * zs[i] for i in range(M)) / sg.sum()
    lvl = lvs[M - 1]
    kl = sum(-.5 * ((1 + lvl[i] - lvl[i].exp()).mean() - (mus[i] ** 2).mean()) for i in range(M))
    l_kl = hyper[0].double() * kl
    l_al = hyper[2].double() * sum(((zs[i] - comb) ** 2).mean() for i in range(M))
    l_f = hyper[3].double() * (comb ** 2).mean()
    l_rec = hyper[1].double() * recp.double().sum()
    total = l_kl + l_al + l_f + sum((dcomb[i].double().sum(0) * comb).sum() for i in range(M))
    grads = torch.autograd.grad(total, mlp + [sg])
    g1_ref = [comb.detach() @ W[i].double().t() + bd[i].double() for i in range(M)]
    # ---- device ----
    d = nv.LatentM()
    d.B, d.L, d.M = B, L, M
    keep = {'sigma': dev(sigma), 'hyper': dev(hyper), 'rec': dev(recp), 'losses': torch.zeros(8, **f32),
            'partials': torch.zeros(20 * nv.load().jamie_max_partials(), **f32), 'dsigma': torch.zeros(M, **f32),
            'comb': torch.zeros(B, L, **f32),
            'colpart': torch.zeros(int(nv.load().jamie_latent_m_colpart_size(B, L)), **f32)}
    keep['losses'][5] = float('inf')
    for i in range(M):
        keep[f'ml{i}'], keep[f'hb{i}'], keep[f'eps_in{i}'] = dev(ml[i]), dev(hb[i]), dev(eps[i])
        keep[f'W{i}'], keep[f'b{i}'], keep[f'dcomb{i}'] = dev(W[i]), dev(bd[i]), dev(dcomb[i])
        keep[f'g1{i}'] = torch.full((B, dims[i]), float('nan'), **f32)
        keep[f'dml{i}'] = torch.zeros(B, 2 * L, **f32)
        keep[f'db{i}'] = dev(prev_db[i]) if acc else torch.zeros(2 * L, **f32)
        keep[f'alias{i}'] = torch.zeros(B, L, **f32)
        for k in ('mu', 'lv', 'z', 'eps'):
            keep[f'{k}{i}'] = torch.zeros(B, L, **f32)
            getattr(d, k)[i] = nv.ptr(keep[f'{k}{i}'])
        d.ml[i], d.head_bias[i], d.eps_in[i] = nv.ptr(keep[f'ml{i}']), nv.ptr(keep[f'hb{i}']), nv.ptr(keep[f'eps_in{i}'])
        d.g1[i], d.dec0_W[i], d.dec0_b[i], d.d[i] = nv.ptr(keep[f'g1{i}']), nv.ptr(keep[f'W{i}']), nv.ptr(keep[f'b{i}']), dims[i]
        d.dcomb[i], d.dml[i], d.dbias_head[i] = nv.ptr(keep[f'dcomb{i}']), nv.ptr(keep[f'dml{i}']), nv.ptr(keep[f'db{i}'])
        d.comb_alias[i] = nv.ptr(keep[f'alias{i}'])
        if with_da2:       # fused tail of the backward launch: the heads' input gradient
            keep[f'Wh{i}'], keep[f'da2{i}'] = dev(Wh[i]), torch.full((B, dims[i]), float('nan'), **f32)
            d.head_W[i], d.da2[i] = nv.ptr(keep[f'Wh{i}']), nv.ptr(keep[f'da2{i}'])
        if bf:
            keep[f'cb{i}'] = torch.zeros(B, L, device='cuda', dtype=torch.bfloat16)
            keep[f'cT{i}'] = torch.zeros(L, B, device='cuda', dtype=torch.bfloat16)
            keep[f'db16{i}'] = torch.zeros(B, 2 * L, device='cuda', dtype=torch.bfloat16)
            keep[f'dT16{i}'] = torch.zeros(2 * L, B, device='cuda', dtype=torch.bfloat16)
            d.comb_bf16[i], d.combT_bf16[i] = nv.ptr(keep[f'cb{i}']), nv.ptr(keep[f'cT{i}'])
            d.dml_bf16[i], d.dmlT_bf16[i] = nv.ptr(keep[f'db16{i}']), nv.ptr(keep[f'dT16{i}'])
    d.ml_nslab, d.ml_slab_stride = nslab, B * 2 * L
    d.sigma, d.hyper, d.partials, d.comb = nv.ptr(keep['sigma']), nv.ptr(keep['hyper']), nv.ptr(keep['partials']), nv.ptr(keep['comb'])
    d.dcomb_nslab, d.dcomb_slab_stride = ndc, B * L
    d.dsigma, d.rec_partials, d.n_rec_partials, d.losses = nv.ptr(keep['dsigma']), nv.ptr(keep['rec']), 5, nv.ptr(keep['losses'])
    keep['ticket'] = torch.zeros(4, dtype=torch.int32, device='cuda')
    d.colpart, d.accumulate, d.rng_stream, d.ticket = nv.ptr(keep['colpart']), int(acc), 100, nv.ptr(keep['ticket'])
    nv.latent_fwd(d, None)
    nv.latent_bwd(d)
    for i in range(M):
        close(keep[f'mu{i}'], mus[i], 1e-5, 1e-6, f'mu{i}')
        close(keep[f'lv{i}'], lvs[i], 1e-5, 1e-6, f'lv{i}')
        close(keep[f'z{i}'], zs[i], 1e-5, 2e-6, f'z{i}')
        close(keep[f'alias{i}'], comb, 1e-5, 2e-6, f'comb alias {i}')
        close(keep[f'g1{i}'], g1_ref[i], 1e-5, 5e-6, f'g1 {i}')
        scale = float(grads[i].abs().max())
        close(keep[f'dml{i}'], grads[i], 2e-4, 2e-6 * scale, f'dml{i}')
        if with_da2:
            want = grads[i] @ Wh[i].double()
            close(keep[f'da2{i}'], want, 2e-4, 2e-6 * float(want.abs().max()), f'da2 {i}')
        want_db = grads[i].sum(0) + (prev_db[i].double() if acc else 0)
        close(keep[f'db{i}'], want_db, 2e-4, 1e-5 * float(want_db.abs().max()), f'head bias grad {i}')
        if bf:
            assert torch.equal(keep[f'cb{i}'], keep['comb'].to(torch.bfloat16)) and torch.equal(keep[f'cT{i}'], keep['comb'].t().to(torch.bfloat16))
            assert torch.equal(keep[f'db16{i}'], keep[f'dml{i}'].to(torch.bfloat16))
            assert torch.equal(keep[f'dT16{i}'], keep[f'dml{i}'].t().to(torch.bfloat16))
    close(keep['comb'], comb, 1e-5, 2e-6, 'comb')
    close(keep['dsigma'], grads[M], 2e-4, 1e-6 * float(grads[M].abs().max()) + 1e-9, 'dsigma')
    assert int(keep['ticket'][0].item()) == 0            # reset by the last workgroup
    ls = keep['losses'].cpu().double()
    close(ls[:4], torch.stack([l_kl, l_rec, l_al, l_f]).detach(), 2e-5, 1e-7, 'losses')
    assert abs(float(ls[4]) - float((l_kl + l_rec + l_al + l_f).detach())) < 2e-5 * abs(float(ls[4])) and ls[5] == ls[4]


def test_latent_rng_eps_is_standard_normal(nv):
    """Without explicit eps the kernel draws N(0,1): check moments and that z = mu + eps*std holds."""
    B, L = 1024, 32
    f32 = dict(device='cuda', dtype=torch.float32)
    d = nv.Latent()
    d.B, d.L = B, L
    keep = {'hyper': torch.ones(16, **f32), 'sigma': torch.ones(2, **f32)}
    for i in range(2):
        keep[f'ml{i}'] = torch.zeros(1, B, 2 * L, **f32)
        keep[f'hb{i}'] = torch.zeros(2 * L, **f32)
        d.ml[i], d.head_bias[i], d.eps_in[i] = nv.ptr(keep[f'ml{i}']), nv.ptr(keep[f'hb{i}']), None
        for k in ('mu', 'lv', 'z', 'eps', 'comb', 'cz'):
            keep[f'{k}{i}'] = torch.zeros(B, L, **f32)
            getattr(d, k)[i] = nv.ptr(keep[f'{k}{i}'])
    d.ml_nslab, d.ml_slab_stride = 1, B * 2 * L
    keep['rsum'], keep['qsum'] = torch.zeros(B, **f32), torch.zeros(B, **f32)
    keep['partials'] = torch.zeros(16 * nv.load().jamie_max_partials(), **f32)
    d.sigma, d.hyper, d.rsum, d.qsum, d.partials = (nv.ptr(keep['sigma']), nv.ptr(keep['hyper']), nv.ptr(keep['rsum']),
                                                    nv.ptr(keep['qsum']), nv.ptr(keep['partials']))
    d.rng_stream = 100
    state = torch.tensor([99, 5, 0, 0], dtype=torch.int64, device='cuda')
    nv.latent_fwd(d, state)
    e0, e1 = keep['eps0'].cpu(), keep['eps1'].cpu()
    assert abs(e0.mean().item()) < 0.02 and abs(e0.std().item() - 1) < 0.02
    assert abs((e0 * e1).mean().item()) < 0.02                       # the two modalities use different streams
    assert abs((e0 ** 4).mean().item() - 3) < 0.2                     # kurtosis of a normal
    close(keep['z0'], e0 * (1 + 1e-7), 1e-6, 1e-6)


def test_latent_m_rng_eps_is_standard_normal(nv):
    """The fused kernel's own draws (one Philox call and one Box-Muller pair per element and pair of modalities): standard
    normal, independent between the modalities of a pair and between pairs, a new draw every step, the same draw in every
    workgroup of a row block (the decoder product of every column chunk must see the same comb), z = mu + eps std."""
    M, B, L, dims = 3, 1024, 32, (520, 264, 136)
    f32 = dict(device='cuda', dtype=torch.float32)
    d = nv.LatentM()
    d.B, d.L, d.M = B, L, M
    keep = {'hyper': torch.ones(16, **f32), 'sigma': torch.ones(M, **f32), 'comb': torch.zeros(B, L, **f32),
            'partials': torch.zeros(20 * nv.load().jamie_max_partials(), **f32)}
    for i in range(M):
        keep[f'ml{i}'], keep[f'hb{i}'] = torch.zeros(1, B, 2 * L, **f32), torch.zeros(2 * L, **f32)
        keep[f'W{i}'], keep[f'b{i}'] = torch.randn(dims[i], L, **f32), torch.zeros(dims[i], **f32)
        keep[f'g1{i}'] = torch.zeros(B, dims[i], **f32)
        d.ml[i], d.head_bias[i], d.eps_in[i] = nv.ptr(keep[f'ml{i}']), nv.ptr(keep[f'hb{i}']), None
        d.g1[i], d.dec0_W[i], d.dec0_b[i], d.d[i] = nv.ptr(keep[f'g1{i}']), nv.ptr(keep[f'W{i}']), nv.ptr(keep[f'b{i}']), dims[i]
        for k in ('mu', 'lv', 'z', 'eps'):
            keep[f'{k}{i}'] = torch.zeros(B, L, **f32)
            getattr(d, k)[i] = nv.ptr(keep[f'{k}{i}'])
    d.ml_nslab, d.ml_slab_stride, d.rng_stream = 1, B * 2 * L, 100
    d.sigma, d.hyper, d.partials, d.comb = nv.ptr(keep['sigma']), nv.ptr(keep['hyper']), nv.ptr(keep['partials']), nv.ptr(keep['comb'])
    state = torch.tensor([99, 5, 0, 0], dtype=torch.int64, device='cuda')
    nv.latent_fwd(d, state)
    e = [keep[f'eps{i}'].cpu().clone() for i in range(M)]
    for i in range(M):
        assert abs(e[i].mean().item()) < 0.02 and abs(e[i].std().item() - 1) < 0.02
        assert abs((e[i] ** 4).mean().item() - 3) < 0.2                    # kurtosis of a normal
        close(keep[f'z{i}'], e[i] * (1 + 1e-7), 1e-6, 1e-6)
        for j in range(i):
            assert abs((e[i] * e[j]).mean().item()) < 0.02                # no correlation within a pair or between pairs
            assert abs((e[i] ** 2 * e[j] ** 2).mean().item() - 1) < 0.05     # ... nor between the magnitudes
    comb = sum(e) / M
    for i in range(M):     # every column chunk's workgroup drew the same eps: its product is comb W^T for the stored comb
        close(keep[f'g1{i}'], comb.double() @ keep[f'W{i}'].cpu().double().t(), 1e-4, 2e-5, f'g1 {i}')
    state[1] += 1
    nv.latent_fwd(d, state)
    assert abs((keep['eps0'].cpu() * e[0]).mean().item()) < 0.02           # a new draw every step


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,gscale_world', [(1000, 1), (4099, 1), (1 << 20, 4)])
def test_clip_adam_matches_oracle(nv, n, gscale_world):
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    params = [p0.clone()]
    opt = orc.Adam(params, 1e-3)
    pd, md, vd = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    state = torch.tensor([0, 0, 0, 0], dtype=torch.int64, device='cuda')
    hyper = torch.zeros(16)
    hyper[8:14] = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1.0, 1.0 / gscale_world])
    hd = dev(hyper)
    nb = nv.optim_blocks(n)
    part = torch.zeros(nb, device='cuda')
    for t in range(5):
        gr = torch.randn(n, generator=g) * (0.01 if t % 2 else 3.0)      # alternate clipped / unclipped
        gsum = gr * gscale_world                                           # what an all-reduce SUM would hold
        gd = dev(gsum)
        nv.grad_sqnorm(gd, part, state)
        nv.clip_adam(pd, gd, md, vd, part, hd, state)
        grads = [gr.clone()]
        orc.clip_grad_norm(grads)
        opt.step(grads)
        close(pd, params[0], rtol=1e-5, atol=2e-6, msg=f'step {t}')
    assert int(state[1].item()) == 5


# ------------------------------------------------------------------------------------------------
# batch assembly
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('d', [24, 70, 2000])
def test_gather_rows(nv, d):
    g = torch.Generator().manual_seed(d)
    src = torch.randn(1000, d, generator=g)
    idx = torch.randint(0, 1000, (300,), generator=g, dtype=torch.int32)
    dst = torch.zeros(300, d, device='cuda')
    nv.gather_rows(dev(src), dev(idx), dst)
    assert torch.equal(dst.cpu(), src[idx.long()])


def test_corr_from_indices(nv):
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 40, (64,), generator=g, dtype=torch.int32)
    corr = torch.zeros(64, 64, device='cuda')
    nv.corr_from_indices(dev(idx), dev(idx), corr)
    close(corr, orc.p_block(None, idx.numpy(), idx.numpy()), 1e-7, 0)


def test_colsum(nv):
    X = torch.randn(3, 130, 50, generator=torch.Generator().manual_seed(8))
    Xd = dev(X)
    out = torch.zeros(50, device='cuda')
    nv.colsum(Xd, 130, 50, 50, out, nslab=3, slab_stride=130 * 50)
    close(out, X.double().sum((0, 1)), 1e-5, 1e-4)           # 390 fp32 terms per column
    nv.colsum(Xd, 130, 50, 50, out, nslab=3, slab_stride=130 * 50, accumulate=True)
    close(out, 2 * X.double().sum((0, 1)), 1e-5, 2e-4)
    # the 16-byte path (rows of whole float4s), grouped launch, row counts around the 8-deep unrolled loop
    g = torch.Generator().manual_seed(9)
    mats = [torch.randn(512, 2000, generator=g), torch.randn(135, 72, generator=g), torch.randn(17, 264, generator=g)]
    outs = [torch.full((m.shape[1],), float('nan'), device='cuda') for m in mats]
    nv.colsum_group([(dev(m), o) for m, o in zip(mats, outs)])
    for m, o in zip(mats, outs):
        close(o, m.double().sum(0), 1e-5, 2e-4)


def test_sampler_without_replacement(nv):
    state = torch.tensor([7, 0, 0, 0], dtype=torch.int64, device='cuda')
    N, B = 100000, 512
    idx = torch.zeros(B, dtype=torch.int32, device='cuda')
    seen = []
    for step in range(20):
        state[1] = step
        nv.sample_indices(idx, N, 0, False, state, 200)
        v = idx.cpu().numpy()
        assert len(set(v.tolist())) == B and v.min() >= 0 and v.max() < N
        seen.append(v.copy())
    again = torch.zeros(B, dtype=torch.int32, device='cuda')
    nv.sample_indices(again, N, 0, False, state, 200)
    assert np.array_equal(again.cpu().numpy(), seen[-1])            # deterministic in (seed, step)
    allv = np.concatenate(seen)
    assert abs(allv.mean() / N - 0.5) < 0.02                          # uniform over [0, N)
    # small N forces collisions: still a permutation-like subset
    idx2 = torch.zeros(48, dtype=torch.int32, device='cuda')
    nv.sample_indices(idx2, 50, 10, False, state, 200)
    v = idx2.cpu().numpy()
    assert len(set(v.tolist())) == 48 and v.min() >= 10 and v.max() < 60
    # with replacement: duplicates allowed, range respected
    idx3 = torch.zeros(512, dtype=torch.int32, device='cuda')
    nv.sample_indices(idx3, 100, 0, True, state, 200)
    v = idx3.cpu().numpy()
    assert v.min() >= 0 and v.max() < 100 and len(set(v.tolist())) < 512


def test_sampler_group_equals_single_draws(nv):
    """jamie_sample_indices_group: several draws in one launch == the same draws launched one by one (the hybrid sampler's
    pair numbers and rows of both modalities, jamie.py:556-583)."""
    state = torch.tensor([5, 17, 0, 0], dtype=torch.int64, device='cuda')
    specs = [(512, 50000, 0, False, 202), (512, 100000, 3, False, 200), (300, 700, 0, True, 201), (48, 50, 10, False, 203)]
    single = [torch.zeros(B, dtype=torch.int32, device='cuda') for B, *_ in specs]
    group = [torch.zeros(B, dtype=torch.int32, device='cuda') for B, *_ in specs]
    for t, (B, N, off, rep, stream) in zip(single, specs):
        nv.sample_indices(t, N, off, rep, state, stream)
    nv.sample_indices_group([nv.sample_args(t, N, off, rep, stream) for t, (B, N, off, rep, stream) in zip(group, specs)], state)
    for a, b in zip(single, group):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# riders: a small launch's work as extra workgroups of a launch that is there anyway
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('what', ['sampler', 'gather'])
def test_clip_adam_riders_equal_their_own_launches(nv, what):
    """jamie_clip_adam_ride: clip + Adam with the next batch's sampler (np.random.choice, jamie.py:556) or its row gather
    + bf16 casts (jamie.py:583) as extra workgroups == the stand-alone launches, bit for bit: parameters, moments, indices
    or gathered rows."""
    n, N, B, d = 3_000_017, 50000, 512, 264
    g = torch.Generator().manual_seed(5)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    hyper = torch.zeros(16)
    hyper[8:14] = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1.0, 1.0])
    hd = dev(hyper)
    data = dev(torch.randn(N, d, generator=g))
    rows = dev(torch.randint(0, N, (B,), generator=g, dtype=torch.int32))
    out = {}
    for ride in (False, True):
        pd, gd = dev(p0), dev(gr)
        md, vd = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
        pb = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
        state = torch.tensor([11, 3, 0, 0], dtype=torch.int64, device='cuda')
        part = torch.zeros(nv.optim_blocks(n), device='cuda')
        idx = torch.zeros(B, dtype=torch.int32, device='cuda')
        x32 = torch.zeros(B, d, device='cuda')
        xbf, xT = torch.zeros(B, d, device='cuda', dtype=torch.bfloat16), torch.zeros(d, B, device='cuda', dtype=torch.bfloat16)
        cast = [nv.cast_problem(data, xbf, xT, rows=rows, dst32=x32)]
        nv.grad_sqnorm(gd, part, state)                     # state[1]: 3 -> 4
        if not ride:
            nv.clip_adam(pd, gd, md, vd, part, hd, state, pb)
            if what == 'sampler':
                nv.sample_indices(idx, N, 7, False, state, 200)
            else:
                nv.cast_transpose(cast)
        elif what == 'sampler':
            nv.clip_adam(pd, gd, md, vd, part, hd, state, pb, sample=nv.sample_args(idx, N, 7, False, 200))
        else:
            nv.clip_adam(pd, gd, md, vd, part, hd, state, pb, casts=cast)
        out[ride] = [t.cpu() for t in (pd, md, vd, pb.float(), idx, x32, xbf.float(), xT.float())]
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)
    if what == 'sampler':
        v = out[True][4].numpy()
        assert len(set(v.tolist())) == B and v.min() >= 7 and v.max() < N + 7
    else:
        assert torch.equal(out[True][5], data.cpu()[rows.cpu().long()])


def test_latent_backward_sampler_rider_draws_the_next_steps_batch(nv):
    """jamie_latent_m_bwd_ex: the extra workgroup draws what jamie_sample_indices draws one step later (step_add = 1: the
    norm kernel has not advanced the counter yet), and the backward launch's own results do not change."""
    M, B, L, N = 2, 512, 32, 100000
    f32 = dict(device='cuda', dtype=torch.float32)
    res = {}
    for ride in (False, True):
        g = torch.Generator().manual_seed(12)
        d = nv.LatentM()
        d.B, d.L, d.M = B, L, M
        keep = {'sigma': torch.ones(M, **f32), 'hyper': torch.ones(16, **f32), 'comb': dev(torch.randn(B, L, generator=g)),
                'partials': torch.zeros(20 * nv.load().jamie_max_partials(), **f32), 'losses': torch.zeros(8, **f32),
                'dsigma': torch.zeros(M, **f32), 'colpart': torch.zeros(int(nv.load().jamie_latent_m_colpart_size(B, L)), **f32),
                'ticket': torch.zeros(4, dtype=torch.int32, device='cuda'), 'rec': torch.zeros(1, **f32)}
        for i in range(M):
            for k in ('mu', 'lv', 'z', 'eps'):
                keep[f'{k}{i}'] = dev(torch.randn(B, L, generator=g) * .3)
                getattr(d, k)[i] = nv.ptr(keep[f'{k}{i}'])
            keep[f'dcomb{i}'], keep[f'dml{i}'] = dev(torch.randn(1, B, L, generator=g) * 1e-3), torch.zeros(B, 2 * L, **f32)
            keep[f'db{i}'] = torch.zeros(2 * L, **f32)
            d.dcomb[i], d.dml[i], d.dbias_head[i] = nv.ptr(keep[f'dcomb{i}']), nv.ptr(keep[f'dml{i}']), nv.ptr(keep[f'db{i}'])
        d.sigma, d.hyper, d.partials, d.comb = nv.ptr(keep['sigma']), nv.ptr(keep['hyper']), nv.ptr(keep['partials']), nv.ptr(keep['comb'])
        d.dcomb_nslab, d.dcomb_slab_stride, d.dsigma, d.losses = 1, B * L, nv.ptr(keep['dsigma']), nv.ptr(keep['losses'])
        d.rec_partials, d.n_rec_partials = nv.ptr(keep['rec']), 1
        d.colpart, d.ticket = nv.ptr(keep['colpart']), nv.ptr(keep['ticket'])
        state = torch.tensor([99, 41, 0, 0], dtype=torch.int64, device='cuda')
        idx = torch.zeros(B, dtype=torch.int32, device='cuda')
        if ride:
            nv.latent_bwd(d, nv.sample_args(idx, N, 0, False, 200, step_add=1), state)
        else:
            nv.latent_bwd(d)
            state[1] += 1
            nv.sample_indices(idx, N, 0, False, state, 200)
        res[ride] = [keep[k].cpu() for k in ('dml0', 'dml1', 'db0', 'db1', 'dsigma', 'losses')] + [idx.cpu()]
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    assert len(set(res[True][-1].tolist())) == B


def test_gemm_bf16_range_norm_rider(nv):
    """jamie_gemm_bf16_ranges: the dW products of a launch (a_tr + b_tr, tile configuration 29) are bit-identical with and
    without riders, the riders' partial sums are the sums of squares of the ranges, block 0 advances the step counter."""
    B, nout, nin = 512, 520, 264
    g = torch.Generator().manual_seed(21)
    dy = dev(torch.randn(B, nout, generator=g)).to(torch.bfloat16)
    a = dev(torch.randn(B, nin, generator=g)).to(torch.bfloat16)
    flat = dev(torch.randn(nout * nin + 10000, generator=g))
    ranges = nv.SqRanges([(nout * nin, 4100), (nout * nin + 4100 + 4, 3), (nout * nin + 5000, 4096)])
    assert ranges.blocks == 4
    outs = {}
    for ride in (False, True):
        dW = torch.zeros(nout, nin, device='cuda')
        tiles = ((nout + 127) // 128) * ((nin + 127) // 128)
        part_dw = torch.zeros(tiles, device='cuda')
        part = torch.full((ranges.blocks,), float('nan'), device='cuda')
        state = torch.tensor([1, 8, 0, 0], dtype=torch.int64, device='cuda')
        prob = [nv.gemm_problem(dy, a, dW, nout, nin, B, nout, nin, nin, partial=part_dw, a_tr=True, b_tr=True, store_nt=True)]
        if ride:
            nv.gemm_bf16(prob, 29, (flat, None, ranges, part, state, None))
        else:
            nv.gemm_bf16(prob, 29)
            nv.grad_sqnorm_ranges(flat, ranges, part, state)
        outs[ride] = (dW.cpu(), part_dw.cpu(), part.cpu(), int(state[1].item()))
    assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])
    assert outs[True][3] == 9 and outs[False][3] == 9
    f = flat.cpu().double()
    want = torch.stack([(f[o:o + n] ** 2).sum() for o, n in ((nout * nin, 4096), (nout * nin + 4096, 4), (nout * nin + 4104, 3),
                                                              (nout * nin + 5000, 4096))])
    close(outs[True][2], want, 1e-5, 1e-6, 'rider partial sums')
    close(outs[False][2], want, 1e-5, 1e-6, 'stand-alone partial sums')
    close(outs[True][0], dy.float().cpu().double().t() @ a.float().cpu().double(), 1e-3, 1e-2, 'dW')


def test_range_norm_takes_longer_chunks_for_long_range_lists(nv):
    """jamie_grad_sqnorm_ranges over more than 128 x 4096 elements (fp32 mode at config 5's dimensions: the skinny head / latent
    matrices are ranges there): the chunk becomes the smallest multiple of 4096 that keeps the count at <= 128; the partial sums
    still add up to the sum of squares of the ranges, the bf16 copies are the rounded ranges, the step counter advances once."""
    g = torch.Generator().manual_seed(8)
    n = 1_600_000
    flat = dev(torch.randn(n, generator=g))
    spec = [(8, 900_000), (900_100, 4), (1_000_000, 500_001)]
    ranges = nv.SqRanges(spec)
    assert ranges.blocks <= 128 and ranges.blocks == sum(-(-ln // 12288) for _, ln in spec)
    part = torch.full((ranges.blocks,), float('nan'), device='cuda')
    state = torch.tensor([1, 8, 0, 0], dtype=torch.int64, device='cuda')
    g16 = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
    nv.grad_sqnorm_ranges(flat, ranges, part, state, g16)
    f = flat.cpu().double()
    want = sum(float((f[o:o + ln] ** 2).sum()) for o, ln in spec)
    assert abs(float(part.double().sum()) - want) <= 1e-6 * want and int(state[1].item()) == 9
    for o, ln in spec:
        assert torch.equal(g16[o:o + ln], flat[o:o + ln].to(torch.bfloat16))
    assert float(g16[:8].float().abs().sum()) == 0.0
    small = nv.SqRanges([(0, 4100), (8200, 3)])                  # short lists keep the 4096-element chunk
    assert small.blocks == 3


# ------------------------------------------------------------------------------------------------
# bf16 compute mode
# ------------------------------------------------------------------------------------------------
def _bf16(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize('R,C,nslab', [(64, 64, 1), (100, 72, 2), (512, 2000, 1), (33, 200, 3)])
def test_cast_transpose(nv, R, C, nslab):
    g = torch.Generator().manual_seed(R + C)
    src = torch.randn(nslab, R, C, generator=g)
    s = dev(src)
    dst = torch.zeros(R, C, dtype=torch.bfloat16, device='cuda')
    dstT = torch.zeros(C, R, dtype=torch.bfloat16, device='cuda')
    nv.cast_transpose([nv.cast_problem(s, dst, dstT, nslab=nslab, slab_stride=R * C)])
    ref = src.sum(0).to(torch.bfloat16)
    assert torch.equal(dst.cpu(), ref)
    assert torch.equal(dstT.cpu(), ref.t())


def test_cast_transpose_with_row_gather(nv):
    """x = data[idx] (jamie.py:583) fused with the bf16 / transposed copies: fp32 rows, bf16 [B, d], bf16 [d, B]."""
    g = torch.Generator().manual_seed(3)
    data = torch.randn(1000, 200, generator=g)
    idx = torch.randint(0, 1000, (130,), generator=g).to(torch.int32)
    x32 = torch.zeros(130, 200, device='cuda')
    xb = torch.zeros(130, 200, dtype=torch.bfloat16, device='cuda')
    xt = torch.zeros(200, 130, dtype=torch.bfloat16, device='cuda')
    nv.cast_transpose([nv.cast_problem(dev(data), xb, xt, rows=dev(idx), dst32=x32)])
    ref = data[idx.long()]
    assert torch.equal(x32.cpu(), ref)
    assert torch.equal(xb.cpu(), ref.to(torch.bfloat16)) and torch.equal(xt.cpu(), ref.to(torch.bfloat16).t())


@pytest.mark.parametrize('M,N,K', [(64, 64, 64), (512, 256, 128), (128, 72, 200), (40, 136, 24), (512, 64, 1000),
                                   (130, 264, 264), (256, 1000, 512)])
@pytest.mark.parametrize('cfg', [-1, 0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28])
def test_gemm_bf16(nv, M, N, K, cfg):
    g = torch.Generator().manual_seed(M + N + K)
    a, w, bias = _bf16(torch.randn(M, K, generator=g)), _bf16(torch.randn(N, K, generator=g)), torch.randn(N, generator=g)
    out = torch.full((M, N), float('nan'), device='cuda')
    A, W, bd = dev(a), dev(w), dev(bias)
    nv.gemm_bf16([nv.gemm_problem(A, W, out, M, N, K, K, K, N, bias=bd)], cfg)
    ref = a.double() @ w.double().t() + bias.double()
    close(out, ref, rtol=1e-5, atol=3e-6 * float(np.sqrt(K)) * 4)     # exact bf16 products, fp32 accumulation


@pytest.mark.parametrize('cfg', [23, 24, 26, 29, 31, 32])
def test_gemm_bf16_stores_survive_the_instructions_behind_them(nv, cfg):
    """Regression (round 4): a 16-byte buffer store with a scalar-register offset had its data registers overwritten by the packed
    multiplies a few instructions behind it (gfx950; hipcc's hazard model exempts that form) -- fixed lanes of fixed rows stored a
    wrong second element.  Integer-valued operands: the product is exact, every element is compared, five launches per shape."""
    for (M, N, K) in ((128, 128, 64), (512, 256, 128), (256, 1000, 512)):
        g = torch.Generator().manual_seed(M + N + K + cfg)
        a = _bf16(torch.randint(-4, 5, (M, K), generator=g).float())
        w = _bf16(torch.randint(-4, 5, (N, K), generator=g).float())
        ref = a.float() @ w.float().t()
        A, W = dev(a), dev(w)
        for rep in range(5):
            out = torch.full((M, N), float('nan'), device='cuda')
            nv.gemm_bf16([nv.gemm_problem(A, W, out, M, N, K, K, K, N)], cfg)
            assert torch.equal(out.cpu(), ref), (cfg, M, N, K, rep, int((out.cpu() != ref).sum()))
    if cfg not in (24, 29):
        return
    # the weight-gradient form: operands stored k-major, bf16 result (8-byte stores with a scalar-register row offset), exact in bf16
    for (M, N, K) in ((256, 384, 64), (2000, 1000, 64)):
        g = torch.Generator().manual_seed(M + N + cfg)
        a = _bf16(torch.randint(-1, 2, (K, M), generator=g).float())
        b = _bf16(torch.randint(-1, 2, (K, N), generator=g).float())
        ref = (a.float().t() @ b.float()).to(torch.bfloat16)
        A, Bm = dev(a), dev(b)
        for rep in range(5):
            out = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
            nv.gemm_bf16([nv.gemm_problem(A, Bm, out, M, N, K, M, N, N, a_tr=True, b_tr=True, store_nt=True, c_bf16=True)], cfg)
            assert torch.equal(out.cpu(), ref), (cfg, M, N, K, rep, int((out.cpu() != ref).sum()))


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (512, 256, 128), (130, 264, 200), (512, 1000, 2000), (40, 136, 24),
                                   (256, 72, 520)])
@pytest.mark.parametrize('cfg', [23, 24, 25])
def test_gemm_bf16_b_stored_k_by_n(nv, M, N, K, cfg):
    """b_tr: C = A [M,K] x B [K,N] with B row-major as stored (the dX product dy W on the weights W [out, in] themselves;
    [k][n] LDS image + ds_read_b64_tr_b16): ragged M / N, partial k-tiles, split-K, grouped with a K-contiguous problem."""
    g = torch.Generator().manual_seed(M + N + K + cfg)
    a, w = _bf16(torch.randn(M, K, generator=g)), _bf16(torch.randn(K, N, generator=g))
    out = torch.full((M, N), float('nan'), device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(a), dev(w), out, M, N, K, K, N, N, b_tr=True)], cfg)
    ref = a.double() @ w.double()
    close(out, ref, rtol=1e-5, atol=3e-6 * float(np.sqrt(K)) * 4)
    # exactly representable values: bit-exact against fp32 torch (catches any swizzle / lane-map slip at every position)
    ai = _bf16(torch.randint(-4, 5, (M, K), generator=g).float())
    wi = _bf16((torch.arange(K * N, dtype=torch.float32).reshape(K, N) % 13) - 6)
    out2 = torch.zeros(M, N, device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(ai), dev(wi), out2, M, N, K, K, N, N, b_tr=True)], cfg)
    assert torch.equal(out2.cpu(), ai.float() @ wi.float())
    if K >= 128:
        sk = 2
        slabs = torch.full((sk, M, N), float('nan'), device='cuda')
        a2, w2 = _bf16(torch.randn(72, 136, generator=g)), _bf16(torch.randn(200, 136, generator=g))
        o2 = torch.zeros(72, 200, device='cuda')
        nv.gemm_bf16([nv.gemm_problem(dev(a), dev(w), slabs, M, N, K, K, N, N, splitk=sk, slab_stride=M * N, b_tr=True),
                      nv.gemm_problem(dev(a2), dev(w2), o2, 72, 200, 136, 136, 136, 200)], cfg)
        close(slabs.sum(0), ref, rtol=1e-5, atol=3e-6 * float(np.sqrt(K)) * 4)
        close(o2, a2.double() @ w2.double().t(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (264, 520, 512), (2000, 1000, 512), (136, 72, 200), (8, 8, 24)])
@pytest.mark.parametrize('cfg', [24, 25])
def test_gemm_bf16_both_operands_stored_k_major(nv, M, N, K, cfg):
    """a_tr + b_tr: C [M,N] = A^T B for A [K,M], B [K,N] row-major (dW = dy^T a on dy [B,out], a [B,in] as the layers
    wrote them), with the per-tile sum-of-squares partials the gradient-norm kernel consumes; accumulate; grouped with
    a dX (b_tr) problem as the backward launch issues them."""
    import math
    g = torch.Generator().manual_seed(M + N + K + cfg)
    a, b = _bf16(torch.randn(K, M, generator=g)), _bf16(torch.randn(K, N, generator=g))
    out = torch.full((M, N), float('nan'), device='cuda')
    part = torch.full((math.ceil(M / 128) * math.ceil(N / 128),), float('nan'), device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(a), dev(b), out, M, N, K, M, N, N, a_tr=True, b_tr=True, partial=part)], cfg)
    ref = a.double().t() @ b.double()
    close(out, ref, rtol=1e-5, atol=3e-6 * float(np.sqrt(K)) * 4)
    close(part.sum(), (out.double() ** 2).sum().cpu(), rtol=1e-5, atol=0)
    ai = _bf16(torch.randint(-4, 5, (K, M), generator=g).float())
    bi = _bf16((torch.arange(K * N, dtype=torch.float32).reshape(K, N) % 13) - 6)
    out2 = torch.ones(M, N, device='cuda')
    w = _bf16(torch.randn(K, 136, generator=g))
    dy = _bf16(torch.randn(72, K, generator=g))
    o3 = torch.zeros(72, 136, device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(dy), dev(w), o3, 72, 136, K, K, 136, 136, b_tr=True),
                  nv.gemm_problem(dev(ai), dev(bi), out2, M, N, K, M, N, N, a_tr=True, b_tr=True, accumulate=True)], cfg)
    assert torch.equal(out2.cpu(), ai.float().t() @ bi.float() + 1.0)
    close(o3, dy.double() @ w.double(), rtol=1e-5, atol=3e-6 * float(np.sqrt(K)) * 4)


def test_gemm_bf16_b_tr_needs_a_128_column_large_tile(nv):
    a, w = dev(_bf16(torch.randn(64, 64))), dev(_bf16(torch.randn(64, 64)))
    out = torch.zeros(64, 64, device='cuda')
    for cfg in (1, 7, 26, 27):
        with pytest.raises(nv.JamieHipError):
            nv.gemm_bf16([nv.gemm_problem(a, w, out, 64, 64, 64, 64, 64, 64, b_tr=True)], cfg)


def test_gemm_bf16_identity_asymmetric(nv):
    n = 96
    B = _bf16(torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251)       # exactly representable
    out = torch.zeros(n, n, device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(_bf16(torch.eye(n))), dev(B.t().contiguous()), out, n, n, n, n, n, n)])
    assert torch.equal(out.cpu(), B.float())


def test_gemm_bf16_splitk_grouped_mse(nv):
    g = torch.Generator().manual_seed(4)
    M, N, K = 200, 48, 1000
    a, w = _bf16(torch.randn(M, K, generator=g)), _bf16(torch.randn(N, K, generator=g))
    a2, w2 = _bf16(torch.randn(64, 72, generator=g)), _bf16(torch.randn(136, 72, generator=g))
    slabs = torch.full((3, M, N), float('nan'), device='cuda')
    o2 = torch.zeros(64, 136, device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(a), dev(w), slabs, M, N, K, K, K, N, splitk=3, slab_stride=M * N),
                  nv.gemm_problem(dev(a2), dev(w2), o2, 64, 136, 72, 72, 72, 136)])
    close(slabs.sum(0), a.double() @ w.double().t(), rtol=1e-5, atol=2e-4)
    close(o2, a2.double() @ w2.double().t(), rtol=1e-5, atol=1e-4)
    import math
    B, d, K = 100, 152, 80
    e2, W, b, X = (_bf16(torch.randn(B, K, generator=g)), _bf16(torch.randn(d, K, generator=g)),
                   torch.randn(d, generator=g), torch.randn(B, d, generator=g))
    bm, bn = nv.gemm_bf16_tile(B, d)
    part = torch.zeros(math.ceil(B / bm) * math.ceil(d / bn), device='cuda')
    out = torch.zeros(B, d, device='cuda')
    nv.gemm_bf16([nv.gemm_problem(dev(e2), dev(W), out, B, d, K, K, K, d, bias=dev(b), epi=nv.EPI_MSE,
                                  aux=(dev(X), None, None, None), aux_ld=d, partial=part, scale=2.0 / (B * d),
                                  pscale=1.0 / (B * d))])
    diff = e2.double() @ W.double().t() + b.double() - X.double()
    close(out, diff * 2.0 / (B * d), rtol=1e-5, atol=1e-7)
    close(part.sum(), (diff ** 2).mean(), rtol=1e-5, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize('R,C,nslab', [(512, 2000, 3), (512, 1000, 2), (100, 72, 1), (65, 130, 2)])
def test_mse_cast_matches_torch(nv, R, C, nslab):
    """jamie_mse_cast: slab sum, (y - x) * scale in fp32 + bf16 + bf16 transposed, per-tile partial sums of squares
    (reference jamie.py:637-641 behind a split-K x_hat GEMM)."""
    torch.manual_seed(R + C)
    y = torch.randn(nslab, R, C, device='cuda')
    x = torch.randn(R, C, device='cuda')
    d = torch.empty(R, C, device='cuda')
    db = torch.empty(R, C, device='cuda', dtype=torch.bfloat16)
    dT = torch.empty(C, R, device='cuda', dtype=torch.bfloat16)
    ntile = ((R + 63) // 64) * ((C + 63) // 64)
    part = torch.zeros(ntile, device='cuda')
    scale, pscale = 2.0 / (R * C), 1.0 / (R * C)
    nv.mse_cast([nv.mse_problem(y, x, d, db, dT, partial=part, scale=scale, pscale=pscale)])
    torch.cuda.synchronize()
    diff = y.sum(0) - x
    assert torch.allclose(d, diff * scale, rtol=1e-6, atol=1e-9)
    assert torch.equal(db, d.to(torch.bfloat16))
    assert torch.equal(dT, d.to(torch.bfloat16).t().contiguous())
    assert abs(part.sum().item() - (diff.double() ** 2).mean().item()) < 1e-5 * (diff.double() ** 2).mean().item()
    # tile order: m fastest (the 64x64 GEMM's fused epilogue)
    t0 = (diff[:64, :64].double() ** 2).sum().item() * pscale
    assert abs(part[0].item() - t0) < 1e-5 * t0
    # without an fp32 output, with the per-tile column sums (round 5: the decoder's output-bias gradient is summed from them)
    db2 = torch.zeros_like(db)
    cp = torch.full(((R + 63) // 64, C), float('nan'), device='cuda')
    nv.mse_cast([nv.mse_problem(y, x, None, db2, None, scale=scale, pscale=pscale, colpart=cp)])
    torch.cuda.synchronize()
    assert torch.equal(db2, db)
    want = torch.stack([d[r0:r0 + 64].double().sum(0) for r0 in range(0, R, 64)])
    assert torch.allclose(cp.double(), want, rtol=1e-5, atol=1e-9) and torch.allclose(cp.sum(0).double(), d.double().sum(0), rtol=1e-5, atol=1e-9)
    with pytest.raises(nv.JamieHipError):
        nv.mse_cast([nv.mse_problem(y, x, None, scale=scale)])


@pytest.mark.parametrize('R,C', [(64, 64), (100, 72), (2000, 1000), (33, 201)])
def test_cast_transpose_bf16_source(nv, R, C):
    """bf16 -> bf16 transposed copy (the weight transposes after the Adam step read its bf16 output)."""
    src = torch.randn(R, C, generator=torch.Generator().manual_seed(R * C)).to(torch.bfloat16).cuda()
    dT = torch.empty(C, R, device='cuda', dtype=torch.bfloat16)
    d = torch.empty(R, C, device='cuda', dtype=torch.bfloat16)
    nv.cast_transpose([nv.cast_problem(src, d, dT)])
    torch.cuda.synchronize()
    assert torch.equal(d, src)
    assert torch.equal(dT, src.t().contiguous())


@pytest.mark.parametrize('N0,N1,B0,B1,dens', [(300, 260, 64, 64, 0.02), (1000, 1000, 128, 96, 0.004), (50, 40, 32, 48, 0.3)])
def test_csr_block_matches_dense_p_block(nv, N0, N1, B0, B1, dens):
    """jamie_csr_block == the oracle's p_block (reference jamie.py:586-589) on the dense matrix: duplicates in the index
    lists, empty rows (divisor 1), row / column offsets of a shard."""
    import scipy.sparse as sp
    rng = np.random.default_rng(N0 + B0)
    Pm = sp.random(N0, N1, density=dens, format='csr', random_state=7, dtype=np.float32)
    Pm.sort_indices()
    dense = torch.from_numpy(Pm.toarray())
    off0, off1 = 5, 3
    idx0 = torch.from_numpy(rng.integers(0, N0 - off0, B0).astype(np.int32))      # with replacement: duplicates
    idx1 = torch.from_numpy(rng.integers(0, N1 - off1, B1).astype(np.int32))
    want = orc.p_block(dense, idx0.long() + off0, idx1.long() + off1)
    out = torch.empty(B0, B1, device='cuda')
    args = [torch.from_numpy(a).cuda() for a in (Pm.indptr.astype(np.int32), Pm.indices.astype(np.int32), Pm.data)]
    nv.csr_block(*args, idx0.cuda(), idx1.cuda(), out, off0, off1, True)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-6, atol=0)
    raw = torch.empty(B0, B1, device='cuda')
    nv.csr_block(*args, idx0.cuda(), idx1.cuda(), raw, off0, off1, False)
    assert torch.equal(raw.cpu(), dense[idx0.long() + off0][:, idx1.long() + off1])


def test_experiments_build_passes_its_own_suite():
    """The kernels of the EXPERIMENTS build (libjamie_hip_exp.so: persistent ring GEMM, fused Linear + BatchNorm launch, register-fed
    skinny products -- built, measured slower, kept out of the product library) against the product kernels they would replace:
    tests/experiments/ in ONE child process that loads that library (JAMIE_LIB).  Skipped where the library was not built."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'jamie_amd', 'libjamie_hip_exp.so')
    if not os.path.exists(lib):
        pytest.skip('experiments library not built (jamie_amd.build.build_experiments())')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'experiments'), '-x', '-q', '-m', 'gpu'],
                       capture_output=True, text=True, env=dict(os.environ, JAMIE_LIB=lib), timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'skipped' not in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_bn_prefetch_rider_changes_nothing(nv):
    """jamie_bn_act_fwd_pf / jamie_bn_act_bwd_pf: the extra workgroups only read the range they are given; outputs are those of
    the plain launches, bit for bit; an unaligned range is refused."""
    g = torch.Generator().manual_seed(3)
    B, N, p = 512, 2000, 0.6
    hs = dev(torch.randn(2, B, N, generator=g))
    gamma, beta = dev(torch.rand(N, generator=g) + .5), dev(torch.randn(N, generator=g))
    da = dev(torch.randn(1, B, N, generator=g))
    state = torch.tensor([5, 1, 0, 0], dtype=torch.int64, device='cuda')
    weights = torch.randn(3_000_001, device='cuda').to(torch.bfloat16)
    outs = []
    for pf in (None, [weights[8:], gamma]):
        h = hs.clone()
        out = torch.zeros(B, N, dtype=torch.bfloat16, device='cuda')
        rm, rv, sm, si = torch.zeros(N, device='cuda'), torch.ones(N, device='cuda'), torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
        pr = nv.BnFwdProblem()
        pr.h, pr.nslab, pr.slab_stride, pr.gamma, pr.beta = nv.ptr(h), 2, B * N, nv.ptr(gamma), nv.ptr(beta)
        pr.running_mean, pr.running_var, pr.save_mean, pr.save_invstd = nv.ptr(rm), nv.ptr(rv), nv.ptr(sm), nv.ptr(si)
        pr.out, pr.mask, pr.out_bf16, pr.B, pr.N, pr.rng_stream = None, None, nv.ptr(out), B, N, 3
        nv.bn_act_fwd([pr], p, state, prefetch=pf)
        d = da.clone()
        dh = torch.zeros(B, N, dtype=torch.bfloat16, device='cuda')
        dg, db, dl = (torch.zeros(N, device='cuda') for _ in range(3))
        pb = nv.BnBwdProblem()
        pb.da, pb.nslab, pb.slab_stride, pb.h, pb.gamma, pb.beta = nv.ptr(d), 1, B * N, nv.ptr(h), nv.ptr(gamma), nv.ptr(beta)
        pb.save_mean, pb.save_invstd, pb.dgamma, pb.dbeta, pb.dbias_lin = nv.ptr(sm), nv.ptr(si), nv.ptr(dg), nv.ptr(db), nv.ptr(dl)
        pb.mask, pb.B, pb.N, pb.rng_stream, pb.accumulate, pb.dh_bf16, pb.skip_f32 = None, B, N, 3, 0, nv.ptr(dh), 1
        nv.bn_act_bwd([pb], p, state, prefetch=pf)
        outs.append((out, rm, rv, sm, si, h[0].clone(), dh, dg, db))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    with pytest.raises(nv.JamieHipError):
        nv.bn_act_fwd([pr], p, state, prefetch=[weights[1:]])        # 2-byte aligned only


# ---- panel layout of the BatchNorm launches' fp32 inputs (round 5; include/jamie_hip.h: JAMIE_PANEL) ----
def _pw():
    from jamie_amd import _native
    return int(_native.load().jamie_panel_width())


def _to_panels(t):
    """[S, B, N] row-major -> the same values in panels of P = jamie_panel_width() columns: flat [S, ceil(N / P) * P * B]."""
    S, B, N = t.shape
    P = _pw()
    npad = (N + P - 1) // P * P
    p = torch.zeros(S, B, npad, dtype=t.dtype, device=t.device)
    p[:, :, :N] = t
    return p.reshape(S, B, npad // P, P).permute(0, 2, 1, 3).contiguous().reshape(S, -1)


def _from_panels(flat, B, N):
    S = flat.shape[0]
    P = _pw()
    npad = (N + P - 1) // P * P
    return flat.reshape(S, npad // P, B, P).permute(0, 2, 1, 3).reshape(S, B, npad)[:, :, :N]


@pytest.mark.parametrize('M,N,K,sk,cfg', [(512, 2000, 1000, 2, 31), (512, 1000, 2000, 3, 32), (512, 264, 512, 1, 29),
                                           (300, 524, 256, 2, 23), (512, 508, 512, 1, 24)])
def test_gemm_bf16_panel_store_equals_row_major_store(nv, M, N, K, sk, cfg):
    """c_panel: every slab of C in panels of 16 columns -- element (m, n) at ((n / 16) * M + m) * 16 + n % 16 -- holds bit for bit
    what the row-major store of the same launch holds (edge tiles in M and N, ragged last panel, bias on slab 0 only)."""
    g = torch.Generator().manual_seed(M + N + K)
    a, w, bias = _bf16(torch.randn(M, K, generator=g)), _bf16(torch.randn(N, K, generator=g)), torch.randn(N, generator=g)
    A, W, bd = dev(a), dev(w), dev(bias)
    ref = torch.full((sk, M, N), float('nan'), device='cuda')
    nv.gemm_bf16([nv.gemm_problem(A, W, ref, M, N, K, K, K, N, bias=bd, splitk=sk, slab_stride=M * N)], cfg)
    P = _pw()
    npad = (N + P - 1) // P * P
    out = torch.full((sk, M * npad), float('nan'), device='cuda')
    nv.gemm_bf16([nv.gemm_problem(A, W, out, M, N, K, K, K, N, bias=bd, splitk=sk, slab_stride=M * npad, c_panel=True)], cfg)
    assert torch.equal(_from_panels(out, M, N), ref)
    # the padding columns of a ragged last panel are never written
    if npad != N:
        pad = out.reshape(sk, npad // P, M, P)[:, -1, :, N % P:]
        assert torch.isnan(pad).all()
    with pytest.raises(nv.JamieHipError):          # small-tile configurations cannot write panels
        nv.gemm_bf16([nv.gemm_problem(A, W, out, M, N, K, K, K, N, bias=bd, splitk=sk, slab_stride=M * npad, c_panel=True)], 7)


@pytest.mark.parametrize('B,N,p,nslab', [(512, 2000, 0.6, 3), (512, 1000, 0.6, 2), (512, 504, 0.0, 1), (256, 72, 0.25, 2), (1024, 136, 0.6, 2)])
def test_bn_act_panel_inputs_equal_row_major_inputs(nv, B, N, p, nslab):
    """BatchNorm forward / backward with `panel` set read their fp32 inputs (split-K slabs, summed pre-activation, upstream
    gradient slabs) from panels of 16 columns and produce bit for bit what the row-major launches produce: bf16 activations,
    statistics, running statistics, the summed pre-activation (written back in panels), bf16 dh, d gamma / d beta / d bias."""
    g = torch.Generator().manual_seed(B + N)
    hs = torch.randn(nslab, B, N, generator=g) * 2 + 0.5
    da_s = torch.randn(nslab, B, N, generator=g)
    gamma, beta = dev(torch.rand(N, generator=g) + .5), dev(torch.randn(N, generator=g))
    state = torch.tensor([77, 5, 0, 0], dtype=torch.int64, device='cuda')

    def run(panel):
        hd = dev(_to_panels(hs)) if panel else dev(hs)
        dad = dev(_to_panels(da_s)) if panel else dev(da_s)
        stride = hd.shape[1] if panel else B * N
        out_bf = torch.zeros(B, N, dtype=torch.bfloat16, device='cuda')
        rm, rv, sm, si = torch.zeros(N, device='cuda'), torch.ones(N, device='cuda'), torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
        pr = nv.BnFwdProblem()
        pr.h, pr.nslab, pr.slab_stride, pr.gamma, pr.beta = nv.ptr(hd), nslab, stride, nv.ptr(gamma), nv.ptr(beta)
        pr.running_mean, pr.running_var, pr.save_mean, pr.save_invstd = nv.ptr(rm), nv.ptr(rv), nv.ptr(sm), nv.ptr(si)
        pr.out, pr.out_bf16, pr.mask, pr.B, pr.N, pr.rng_stream, pr.panel = None, nv.ptr(out_bf), None, B, N, 3, int(panel)
        nv.bn_act_fwd([pr], p, state)
        dh_bf = torch.zeros(B, N, dtype=torch.bfloat16, device='cuda')
        dg, db, dl = (torch.zeros(N, device='cuda') for _ in range(3))
        pb = nv.BnBwdProblem()
        pb.da, pb.nslab, pb.slab_stride = nv.ptr(dad), nslab, stride
        pb.h, pb.gamma, pb.beta, pb.save_mean, pb.save_invstd = nv.ptr(hd), nv.ptr(gamma), nv.ptr(beta), nv.ptr(sm), nv.ptr(si)
        pb.dgamma, pb.dbeta, pb.dbias_lin, pb.mask = nv.ptr(dg), nv.ptr(db), nv.ptr(dl), None
        pb.B, pb.N, pb.rng_stream, pb.accumulate, pb.dh_bf16, pb.skip_f32, pb.panel = B, N, 3, 0, nv.ptr(dh_bf), 1, int(panel)
        nv.bn_act_bwd([pb], p, state)
        torch.cuda.synchronize()
        h0 = _from_panels(hd[:1], B, N)[0] if panel else hd[0]
        return out_bf, rm, rv, sm, si, h0.clone(), dh_bf, dg, db, dl
    a, b = run(False), run(True)
    # (the panel launches may take another strip width -- 8 columns where that halves the busiest CU's columns -- so the column
    #  statistics are summed in another order: equal to fp32 rounding, the bf16 outputs equal except where a value sits on a
    #  rounding boundary)
    for x, y, name in zip(a, b, ('out', 'running_mean', 'running_var', 'save_mean', 'save_invstd', 'h sum', 'dh', 'dgamma', 'dbeta', 'dbias')):
        if x.dtype == torch.bfloat16:
            off = x != y
            assert float(off.float().mean()) < 2e-3, (name, float(off.float().mean()))
            assert float((x.float() - y.float()).abs().max()) <= 2 ** -7 * float(x.float().abs().max()), name
        elif name == 'h sum':
            assert torch.equal(x, y), name                      # (the slab sums are per element: no reduction order involved)
        elif name == 'dbias':
            assert float((x - y).abs().max()) < 1e-3, name      # (column sums of dh: mathematically zero)
        else:
            close(x, y.cpu().double(), rtol=2e-5, atol=2e-5 if name in ('dgamma', 'dbeta') else 1e-6)
    close(a[5], hs.sum(0), rtol=1e-6, atol=1e-6)
    # a panel-layout backward launch cannot write the fp32 dh in place (its input is in another layout)
    pb = nv.BnBwdProblem()
    t = torch.zeros(1, B * ((N + _pw() - 1) // _pw() * _pw()), device='cuda')
    pb.da, pb.nslab, pb.slab_stride, pb.h = nv.ptr(t), 1, t.shape[1], nv.ptr(t)
    pb.gamma, pb.beta, pb.save_mean, pb.save_invstd = nv.ptr(gamma), nv.ptr(beta), nv.ptr(gamma), nv.ptr(gamma)
    pb.dgamma, pb.dbeta, pb.B, pb.N, pb.panel = nv.ptr(torch.zeros(N, device='cuda')), nv.ptr(torch.zeros(N, device='cuda')), B, N, 1
    with pytest.raises(nv.JamieHipError):
        nv.bn_act_bwd([pb], 0.0, state)
