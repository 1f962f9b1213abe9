"""Correspondence stage on the MI355X: `Prime_Dual` through the C ABI against the
reference's own outputs (tests/golden/pd*.npz, tools/make_goldens_pd.py) and against the CPU oracle.
Run on the GPU box:  pytest -m gpu"""
import ast
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import jamie_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def jam():
    import jamie_amd
    from jamie_amd import _native
    _native.require_gpu()
    return jamie_amd


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


@pytest.mark.parametrize('name', ['pd1_delay0', 'pd2_delay'])
def test_prime_dual_vs_reference_golden(jam, name):
    """F after all iterations against the reference's F (fp32 iteration: same accuracy class, not bit-equal)."""
    from jamie_amd.correspondence import prime_dual
    g = np.load(os.path.join(GOLD, name + '.npz'))
    m = ast.literal_eval(str(g['meta']))
    F = prime_dual([g['Kx'], g['Ky']], m['dx'], m['dy'], epoch_pd=m['epoch_pd'], rho=m['rho'], epsilon=m['epsilon'],
                   delay=m['delay'], verbose=False)
    assert F.shape == g['F'].shape and F.dtype == np.float32
    np.testing.assert_allclose(F, g['F'], rtol=2e-3, atol=2e-6)
    assert _rel(F, g['F']) < 2e-4


@pytest.mark.parametrize('m,n,delay,eps', [(130, 117, 0, 1e-3), (256, 200, 25, 5e-3), (65, 300, 3, 1e-2)])
def test_prime_dual_vs_oracle(jam, m, n, delay, eps):
    """Ragged shapes (n % 4 != 0 takes the scalar path; several column / row tiles), with and without a delay;
    the scaling factor `a` is followed iteration by iteration."""
    from jamie_amd.correspondence import PrimeDual
    from sklearn.metrics import pairwise_distances
    rng = np.random.default_rng(m + n)
    Z = rng.standard_normal((max(m, n), 5))
    Kx = pairwise_distances(Z[:m] @ rng.standard_normal((5, 20)))
    Ky = pairwise_distances(Z[:n] @ rng.standard_normal((5, 14)))
    iters = 60
    hist = []
    want = orc.prime_dual(Kx, Ky, 20, 14, iters, 10, eps, delay, history=hist)
    pd = PrimeDual(Kx, Ky, 20, 14, rho=10, epsilon=eps, delay=delay)
    got_a = []
    for _ in range(iters):
        pd.step()
        got_a.append(float(pd.alpha.item()))
    np.testing.assert_allclose(got_a, hist, rtol=2e-4)
    F = pd.F.cpu().numpy()
    np.testing.assert_allclose(F, want, rtol=2e-3, atol=2e-6)
    assert _rel(F, want) < 2e-4
    # the device-side row / column sums are those of the F it holds
    np.testing.assert_allclose(pd.rowsum.cpu().numpy(), F.sum(1), rtol=1e-5)
    np.testing.assert_allclose(pd.colsum.cpu().numpy(), F.sum(0), rtol=1e-5)
    err, a = pd.error()
    assert np.isfinite(err) and abs(a - hist[-1]) < 2e-4 * abs(hist[-1])


def test_prime_dual_large_products_on_the_bf16_pipe(jam):
    """N >= 1024: the four products of an iteration run on the bf16 matrix pipe (three-piece cuts of every fp32 element,
    gemm_f32.hip configuration 20; 21 from 3072).  40 iterations at 1280 x 1152 against the same iterations on the fp32 pipe
    (gemm_cfg=17, the path the small-size tests pin to the oracle and the reference's goldens): the scaling factor iteration by
    iteration and the final F."""
    from jamie_amd.correspondence import PrimeDual
    m, n = 1280, 1152
    g = torch.Generator(device='cuda').manual_seed(3)
    X, Y = torch.randn(m, 24, generator=g, device='cuda'), torch.randn(n, 16, generator=g, device='cuda')
    Kx, Ky = torch.cdist(X, X), torch.cdist(Y, Y)
    runs = {}
    for cfg in (None, 17):
        pd = PrimeDual(Kx, Ky, 24, 16, rho=10, epsilon=1e-3, delay=5, device='cuda', gemm_cfg=cfg)
        assert pd.gemm_cfg == (20 if cfg is None else 17)
        alphas = []
        for _ in range(40):
            pd.step()
            alphas.append(float(pd.alpha.item()))
        runs[cfg] = (np.array(alphas), pd.F.double().cpu().numpy())
    np.testing.assert_allclose(runs[None][0], runs[17][0], rtol=2e-5)
    assert np.isfinite(runs[None][1]).all()
    assert _rel(runs[None][1], runs[17][1]) < 1e-4
    assert PrimeDual(torch.zeros(4096, 4096, device='cuda'), torch.zeros(4096, 4096, device='cuda'), 8, 8, device='cuda').gemm_cfg == 21


def test_prime_dual_is_deterministic(jam):
    from jamie_amd.correspondence import prime_dual
    g = np.load(os.path.join(GOLD, 'pd1_delay0.npz'))
    a = prime_dual([g['Kx'], g['Ky']], 30, 20, epoch_pd=40, verbose=False)
    b = prime_dual([g['Kx'], g['Ky']], 30, 20, epoch_pd=40, verbose=False)
    assert np.array_equal(a, b)


def test_prime_dual_1x1_escape(jam):
    from jamie_amd.correspondence import prime_dual
    with pytest.warns(UserWarning):
        F = prime_dual([np.zeros((1, 1)), np.zeros((1, 1))], 3, 2, verbose=False)
    assert F.shape == (1, 1) and F[0, 0] == 1


def test_facade_stage_a_b_vs_reference_pipeline(jam):
    """compute_distances -> match through the facade reproduces the reference's stage A/B output (pd3 fixture:
    euclidean distances of both modalities, then Prime_Dual), and a whole fit with use_f_tilde=True runs on it."""
    g = np.load(os.path.join(GOLD, 'pd3_pipeline.npz'))
    m = ast.literal_eval(str(g['meta']))
    with contextlib.redirect_stdout(io.StringIO()):
        jm = jam.JAMIE(distance_mode='euclidean', epoch_pd=m['epoch_pd'], output_dim=4, batch_size=56, epoch_DNN=12,
                       min_epochs=5, pca_dim=None, use_f_tilde=True, log_DNN=10 ** 9, log_pd=50)
        emb = jm.fit_transform(dataset=[g['X'], g['Y']])
    for got, want in zip(jm.dist, (g['dist0'], g['dist1'])):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    F = np.asarray(jm.match_result[0])
    np.testing.assert_allclose(F, g['F'], rtol=2e-3, atol=2e-6)
    assert len(emb) == 2 and emb[0].shape == (56, 4) and np.isfinite(emb[0]).all() and np.isfinite(emb[1]).all()
    assert np.isfinite(jm.loss_history['F']).all() and jm.loss_history['F'][0] > 0
