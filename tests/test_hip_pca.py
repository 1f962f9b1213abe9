"""Device PCA (`jamie_amd/pca.py`) against sklearn, the library the reference calls when `pca_dim`
is set (its default; reference jamie/jamie.py:436-457), and `preclass(sample, pca=pca)` (utilities.py:654-678).
PCA components are defined up to what the spectrum separates, so the checks are: explained variance (1e-3 of FULL PCA),
the components where singular values are separated (signed: same `svd_flip` rule as sklearn), the scores, the round trip,
and the facade with `preprocess='device'` against `preprocess='host'`.
Run on the MI355X box:  pytest -m gpu"""
import contextlib
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def jam():
    import jamie_amd
    from jamie_amd import _native
    _native.require_gpu()
    return jamie_amd


def _spectrum_data(N, d, rank, seed, decay=0.8, noise=0.02):
    """Cells x features with geometrically separated singular values on `rank` directions + a little noise."""
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((N, rank)))
    V, _ = np.linalg.qr(rng.standard_normal((d, rank)))
    s = 50.0 * decay ** np.arange(rank)
    return (U * s) @ V.T * np.sqrt(N) / 10 + noise * rng.standard_normal((N, d)) + rng.standard_normal(d) * 3.0


@pytest.mark.parametrize('N,d,k,decay', [(3000, 400, 24, 0.8), (300, 2000, 16, 0.8), (20000, 1000, 64, 0.9),
                                         (4000, 300, 40, 0.7)])
def test_device_pca_matches_sklearn(jam, N, d, k, decay):
    """Neighbouring singular values in ratio `decay` (separated components); the last case spreads the top-k singular values
    over 6 decades (explained variances over 12: beyond what an fp32 Gram matrix resolves, so the [N, l] bases take the
    float64 host-QR fallback) -- the explained variances still match FULL float64 PCA to 1e-3 down to the noise floor."""
    from sklearn.decomposition import PCA
    from jamie_amd.pca import DevicePCA
    X = _spectrum_data(N, d, rank=k + 8, seed=N + d, decay=decay, noise=0.002)
    full = PCA(n_components=k, svd_solver='full').fit(X)
    dp = DevicePCA(k, random_state=0)
    scores = dp.fit_transform_device(torch.from_numpy(X)).cpu().numpy()
    assert dp.components_.shape == (k, d) and scores.shape == (N, k) and dp.n_components_ == k
    # explained variance within 1e-3 of FULL PCA, ratios alike
    # (fp32 products resolve a variance to ~1e-7 of the LARGEST one: the atol; only the 6-decade case gets there)
    top = full.explained_variance_[0]
    np.testing.assert_allclose(dp.explained_variance_, full.explained_variance_, rtol=1e-3, atol=1e-7 * top)
    np.testing.assert_allclose(dp.explained_variance_ratio_, full.explained_variance_ratio_, rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(dp.singular_values_ ** 2, full.singular_values_ ** 2, rtol=1e-3,
                               atol=1e-7 * full.singular_values_[0] ** 2)
    np.testing.assert_allclose(dp.mean_, full.mean_, rtol=1e-6, atol=1e-8)
    # components: separated spectrum (ratio 0.8 between neighbours) -> the vectors themselves agree, signs included
    sep = full.explained_variance_ > 1e-5 * top            # components above the fp32 floor / the noise floor
    cos = np.sum(dp.components_ * full.components_, axis=1)
    assert cos[sep].min() > 0.999 and sep.sum() >= min(k, 16), (cos[sep].min(), sep.sum())
    np.testing.assert_allclose(scores[:, sep], full.transform(X)[:, sep], rtol=2e-3, atol=2e-3 * np.abs(scores).max())
    # the same algorithm in sklearn (randomized, same Omega: RandomState(0).normal(size=(d, k + 10)))
    rnd = PCA(n_components=k, svd_solver='randomized', random_state=0).fit(X)
    np.testing.assert_allclose(dp.explained_variance_, rnd.explained_variance_, rtol=1e-3, atol=1e-7 * top)
    assert np.sum(dp.components_ * rnd.components_, axis=1)[sep].min() > 0.999
    # orthonormal components
    np.testing.assert_allclose(dp.components_ @ dp.components_.T, np.eye(k), atol=1e-5)
    # transform / inverse_transform: numpy path (few rows) and device path (many rows) agree with sklearn's
    few = X[:100]
    np.testing.assert_allclose(dp.transform(few), (few - dp.mean_) @ dp.components_.T, rtol=1e-9, atol=1e-9)
    many = np.concatenate([X] * (2048 // N + 1))[:max(2048, min(N, 4096))]
    zt = dp.transform(many)
    np.testing.assert_allclose(zt, (many - dp.mean_) @ dp.components_.T, rtol=1e-3, atol=1e-3 * np.abs(zt).max())
    back = dp.inverse_transform(zt)
    np.testing.assert_allclose(back, zt @ dp.components_ + dp.mean_, rtol=1e-3, atol=1e-3 * np.abs(back).max())


def test_device_pca_consumes_the_global_rng_like_sklearn(jam):
    """random_state=None is numpy's global RandomState, as in sklearn (`check_random_state(None)`), and the fit draws exactly
    one normal(size=(d, k + 10)): a seeded reference-style run sees the same sampler stream after preprocessing."""
    from jamie_amd.pca import DevicePCA
    X = _spectrum_data(600, 120, rank=20, seed=3)
    np.random.seed(5)
    DevicePCA(12).fit(X)
    after = np.random.rand(3)
    np.random.seed(5)
    np.random.normal(size=(120, 22))
    np.testing.assert_array_equal(after, np.random.rand(3))


def test_global_standardise_is_preclass_axis_none(jam):
    from jamie_amd.pca import global_standardise
    from jamie_amd.utilities import preclass
    rng = np.random.default_rng(0)
    S = (rng.standard_normal((5000, 37)) * np.linspace(0.1, 9, 37) + 0.3).astype(np.float32)
    out, m, s = global_standardise(torch.from_numpy(S).cuda())
    ref = preclass(S.astype(np.float64), axis=None)
    assert abs(m - float(ref.mean)) < 1e-9 + 1e-7 * abs(float(ref.mean)) and abs(s - float(ref.std)) < 1e-7 * float(ref.std)
    np.testing.assert_allclose(out.cpu().numpy(), ref.transform(S.astype(np.float64)), rtol=1e-5, atol=1e-6)


def test_facade_device_pca_preprocessing(jam):
    """`JAMIE(pca_dim=[k0, k1], preprocess='device')`: the fitted preprocessing equals the host path's (sklearn PCA +
    `preclass(axis=None)`) up to the sign of a component, the model trains on the scores, `transform` / `modal_predict`
    map new raw cells through it (imputed matrices come back in the raw feature space, float64) and `save_model` /
    `load_model` carry the fitted PCA."""
    rng = np.random.default_rng(11)
    N, dims, k = 1500, (300, 200), (12, 10)
    Z = rng.standard_normal((N, 6)) * np.array([9, 7, 5, 4, 3, 2.])
    data = [Z @ rng.standard_normal((6, d)) + 0.3 * rng.standard_normal((N, d)) * np.linspace(2, 0.1, d) for d in dims]
    runs = {}
    for mode in ('host', 'device'):
        np.random.seed(1)
        jm = jam.JAMIE(output_dim=6, batch_size=128, epoch_DNN=4, min_epochs=2, pca_dim=list(k), use_f_tilde=False,
                       log_DNN=10 ** 9, preprocess=mode)
        with contextlib.redirect_stdout(io.StringIO()):
            emb = jm.fit_transform(dataset=[d.copy() for d in data])
        assert [e.shape for e in emb] == [(N, 6), (N, 6)] and all(np.isfinite(e).all() for e in emb)
        assert jm.model.input_dim == list(k)
        runs[mode] = jm
    for i in range(2):
        th = runs['host'].model.preprocessing[i](data[i].copy())
        td = runs['device'].model.preprocessing[i](data[i].copy())
        assert th.shape == td.shape == (N, k[i])
        # the 6 signal directions are separated; the remaining components span noise: compare the leading columns (sign-free)
        for c in range(5):
            r = abs(np.corrcoef(th[:, c], td[:, c])[0, 1])
            assert r > 0.999, (i, c, r)
        np.testing.assert_allclose(np.abs(td).std(), np.abs(th).std(), rtol=2e-2)
        assert abs(td.std() - 1.0) < 1e-3 and abs(td.mean()) < 1e-3          # preclass(axis=None): ONE mean / std
    jm = runs['device']
    imp = jm.modal_predict(data[0], 0)
    assert imp.shape == (N, dims[1]) and imp.dtype == np.float64 and np.isfinite(imp).all()
    tr = jm.transform(data)
    buf = io.BytesIO()
    jm.save_model(buf)
    buf.seek(0)
    jm2 = jam.JAMIE(output_dim=6, use_f_tilde=False)
    jm2.load_model(buf)
    np.testing.assert_array_equal(jm2.transform(data)[0], tr[0])
    np.testing.assert_allclose(jm2.modal_predict(data[0], 0), imp, rtol=1e-12)


def test_device_pca_constant_matrix_gives_finite_zero_scores(jam):
    """A constant modality (zero variance after centring): the sketch's Gram matrix is all zero; the scores are finite zeros
    like sklearn's, not V / sqrt(0)."""
    from jamie_amd.pca import DevicePCA
    X = np.full((600, 120), 3.25, dtype=np.float32)
    dp = DevicePCA(8, random_state=0)
    scores = dp.fit_transform_device(torch.from_numpy(X)).cpu().numpy()
    assert scores.shape == (600, 8) and np.isfinite(scores).all() and np.abs(scores).max() < 1e-4
    assert np.isfinite(dp.components_).all() and np.isfinite(dp.explained_variance_).all()
    assert np.abs(dp.explained_variance_).max() < 1e-8
    np.testing.assert_allclose(dp.mean_, 3.25)
