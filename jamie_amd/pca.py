"""PCA of a cells x features matrix on the MI355X: the device counterpart of the `sklearn.decomposition.PCA(n_components)`
that the reference fits per modality when `pca_dim` is set -- its DEFAULT (`pca_dim=2*[512]`, reference jamie/jamie.py:50,
436-457) and the bulk of its 'Setup' phase (47-291 s on the high-dimensional datasets, time-and-memory.ipynb).
SURVEY.md §8(f) rank 4.

Algorithm: the randomized range finder of Halko, Martinsson & Tropp (2011), which is also what sklearn's PCA runs for
these shapes (`svd_solver='randomized'`: `n_oversamples=10`, `n_iter=4` or 7, LU-normalised power iterations,
`svd_flip(u_based_decision=False)`).  Every product that touches the N x d matrix runs on the exact-fp32 MFMA GEMM of
this library (`jamie_gemm_f32`: NN, TN and NT layouts, split-K for the short-and-wide outputs); the l x l algebra
(l = n_components + 10) is done on the host in float64:

    Xc      = X - mean                                          jamie_col_stats + jamie_standardise (sd = 1)
    Y       = Xc Q          [N, l]     NN                       Q = Omega ~ N(0, 1) [d, l] first
    orthonormalise: G = Y^T Y [l, l]   TN, split-K;  host: G = V diag(w) V^T;  Y <- Y V diag(w)^-1/2   NN
    Z       = Xc^T Y        [d, l]     TN, split-K;  orthonormalise the same way; n_iter times
    B       = Q^T Xc        [l, d]     TN, split-K;  host: B = U S V^T (float64)
    components = V^T[:k] (sign: largest |entry| of every row positive), explained variance = S^2 / (N - 1)
    scores  = Xc components^T [N, k]   NT                       (the matrix the model trains on)

The [N, l] bases are orthonormalised by Gram / eigen whitening (device products, float64 l x l algebra; the final basis
twice, the fp32 analogue of CholeskyQR2: orthonormal to ~1e-6), the small [d, l] bases by a float64 Householder QR on the
host; a sketch whose singular values spread over more than ~2.5 decades (cond^2 beyond fp32) falls back to the host QR for
the [N, l] side too.  What fp32 products resolve: explained variances down to ~1e-6 of the largest to 1e-3 relative
(tests/test_hip_pca.py).  Rows are processed in blocks of < 4 GiB so that the GEMM keeps its buffer-load path.
The fitted object keeps `mean_`, `components_`, ... as float64 numpy arrays (picklable: `save_model` stores the
preprocessing callables, jamie.py:967-968) and offers sklearn's `transform` / `inverse_transform`.
"""
import math

import numpy as np
import torch

from . import _native as nv

_ROW_BLOCK_BYTES = 3 << 30


def _splitk(M, N, K, bm=128, bn=128):
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    return int(max(1, min(64, math.ceil(1024 / tiles), K // 2048)))


def _cfg(M, N, K):
    return 12 if (M >= 512 and N >= 256 and K >= 256) else -1


def _sum_slabs(slabs, out):
    """out [R, C] fp32 = sum of the split-K slabs [S, R, C] (jamie_cast_transpose's slab sum with an fp32 destination)."""
    S, R, Cc = slabs.shape
    nv.cast_transpose([nv.cast_problem(slabs, None, None, nslab=S, slab_stride=R * Cc, dst32=out)])
    return out


def _row_blocks(n_rows, row_bytes):
    step = max(1, min(n_rows, _ROW_BLOCK_BYTES // max(1, row_bytes)))
    return [(lo, min(n_rows, lo + step)) for lo in range(0, n_rows, step)]


def mm_nn(A, B, out=None, bias=None):
    """A [M, K] @ B [K, N] (+ bias[N]) on the fp32 MFMA GEMM, A in row blocks."""
    M, K = A.shape
    N = B.shape[1]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32) if out is None else out
    for lo, hi in _row_blocks(M, K * 4):
        nv.gemm([nv.gemm_problem(A[lo:hi], B, out[lo:hi], hi - lo, N, K, A.stride(0), B.stride(0), out.stride(0), bias=bias)],
                nv.NN, _cfg(hi - lo, N, K))
    return out


def mm_nt(A, B, out=None):
    """A [M, K] @ B[N, K]^T."""
    M, K = A.shape
    N = B.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32) if out is None else out
    for lo, hi in _row_blocks(M, K * 4):
        nv.gemm([nv.gemm_problem(A[lo:hi], B, out[lo:hi], hi - lo, N, K, A.stride(0), B.stride(0), out.stride(0))],
                nv.NT, _cfg(hi - lo, N, K))
    return out


def mm_tn(A, B):
    """A[K, M]^T @ B [K, N] with K (the cells) long: split-K slabs per row block of K, summed by one launch."""
    K, M = A.shape
    N = B.shape[1]
    blocks = _row_blocks(K, max(M, N) * 4)
    sk = [_splitk(M, N, hi - lo) for lo, hi in blocks]
    slabs = torch.empty(sum(sk), M, N, device=A.device, dtype=torch.float32)
    s0 = 0
    for (lo, hi), s in zip(blocks, sk):
        nv.gemm([nv.gemm_problem(A[lo:hi], B[lo:hi], slabs[s0:s0 + s], M, N, hi - lo, A.stride(0), B.stride(0), N,
                                 splitk=s, slab_stride=M * N)], nv.TN, _cfg(M, N, hi - lo))
        s0 += s
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    return _sum_slabs(slabs, out) if slabs.shape[0] > 1 else out.copy_(slabs[0])


def _host_qr(Y):
    """Orthonormal basis of the columns of Y [n, l] by Householder QR in float64 on the host."""
    q, _ = np.linalg.qr(Y.double().cpu().numpy())
    return torch.from_numpy(np.ascontiguousarray(q.astype(np.float32))).to(Y.device)


def _whiten(Y, rounds=1, cond_limit=1e5):
    """Y [n, l] -> Y T with T = V diag(w)^-1/2 from the eigen-decomposition of the Gram matrix Y^T Y (device product,
    float64 host algebra): orthonormal columns to ~1e-6 after two rounds (the fp32 analogue of CholeskyQR2) as long as
    cond(Y)^2 = w_max / w_min stays well inside fp32; beyond `cond_limit` the Gram matrix no longer resolves the small
    directions and the basis comes from a float64 Householder QR on the host instead (n l^2 host FLOP: the rare,
    ill-conditioned case -- singular values spread over more than ~2.5 decades inside the sketch)."""
    for _ in range(rounds):
        G = mm_tn(Y, Y).double().cpu().numpy()
        w, V = np.linalg.eigh((G + G.T) * 0.5)
        # (w.max() <= 0: an all-zero sketch -- a constant modality / zero-variance features after centring -- would give
        #  V / sqrt(0) = inf / NaN scores; the host QR returns a finite basis, as sklearn does)
        if w.max() <= 0 or not np.isfinite(w).all() or w.min() * cond_limit < w.max():
            return _host_qr(Y)
        T = torch.from_numpy((V / np.sqrt(w)).astype(np.float32)).to(Y.device)
        Y = mm_nn(Y, T)
    return Y


def _orth_small(Z):
    """Feature-side basis [d, l]: d l^2 is small next to the products with the cells, so it is always the exact host QR
    unless the matrix is very tall (hundreds of thousands of features), where the device whitening takes over."""
    d, ell = Z.shape
    return _host_qr(Z) if d * ell * ell <= 4e10 else _whiten(Z, rounds=2)


def _center(X, mean=None):
    """(X - column mean) as fp32 on the device; the mean in float64."""
    N, d = X.shape
    f64 = int(X.dtype == torch.float64)
    if mean is None:
        R = int(max(1, min(256, (N + 2047) // 2048)))
        part = torch.empty(R * d, dtype=torch.float64, device=X.device)
        mean = torch.empty(d, dtype=torch.float64, device=X.device)
        sd = torch.empty(d, dtype=torch.float64, device=X.device)
        nv._call('jamie_col_stats', nv.ptr(X), f64, N, d, d, nv.ptr(part), R, nv.ptr(mean), nv.ptr(sd), nv._stream())
    else:
        sd = None
    ones = torch.ones(d, dtype=torch.float64, device=X.device)
    out = torch.empty(N, d, dtype=torch.float32, device=X.device)
    nv._call('jamie_standardise', nv.ptr(X), f64, N, d, d, nv.ptr(mean), nv.ptr(ones), nv.ptr(out), nv._stream())
    return out, mean, sd


def _as_device_matrix(X, device):
    X = torch.as_tensor(np.ascontiguousarray(X)) if not torch.is_tensor(X) else X
    if X.dtype not in (torch.float32, torch.float64):
        X = X.to(torch.float64)
    return X.to(device).contiguous()


class DevicePCA:
    """`sklearn.decomposition.PCA(n_components=k)` fitted on the GPU (see the module docstring).

    Attributes after `fit` (float64 numpy, sklearn's names): `mean_`, `components_` [k, d], `explained_variance_`,
    `explained_variance_ratio_`, `singular_values_`, `n_components_`, `n_samples_`, `n_features_in_`.
    `fit_transform_device(X)` returns the scores [N, k] as an fp32 GPU tensor (what the training loop gathers from)."""

    def __init__(self, n_components, n_oversamples=10, n_iter='auto', random_state=None, device='cuda'):
        self.n_components = int(n_components)
        self.n_oversamples = int(n_oversamples)
        self.n_iter = n_iter
        self.random_state = random_state
        self.device = str(device)

    # ---- fit ----
    def fit_transform_device(self, X):
        nv.require_gpu()
        dev = torch.device(self.device)
        X = _as_device_matrix(X, dev)
        N, d = X.shape
        k = min(self.n_components, N, d)
        ell = min(k + self.n_oversamples, N, d)
        n_iter = self.n_iter
        if n_iter == 'auto':                                   # sklearn.utils.extmath.randomized_svd
            n_iter = 7 if k < 0.1 * min(N, d) else 4
        Xc, mean, sd = _center(X)
        del X
        # sklearn's check_random_state: None = numpy's GLOBAL RandomState (so a seeded run consumes the same draws as the
        # reference's PCA(n_components) does: one normal(size=(d, k + 10)) call)
        rs = self.random_state
        if rs is None:
            rs = np.random.mtrand._rand
        elif not isinstance(rs, np.random.RandomState):
            rs = np.random.RandomState(rs)
        Q = torch.from_numpy(rs.normal(size=(d, ell)).astype(np.float32)).to(dev)
        for _ in range(int(n_iter)):                           # power iterations, re-conditioned after every product
            Q = _whiten(mm_nn(Xc, Q))                          # [N, l]
            Q = _orth_small(mm_tn(Xc, Q))                      # [d, l]
        Q = _whiten(mm_nn(Xc, Q), rounds=2)                    # orthonormal basis of the range of Xc, [N, l]
        B = mm_tn(Q, Xc).double().cpu().numpy()                # [l, d]
        _, S, Vt = np.linalg.svd(B, full_matrices=False)
        Vt, S = Vt[:k], S[:k]
        idx = np.argmax(np.abs(Vt), axis=1)                    # svd_flip(u_based_decision=False)
        Vt = Vt * np.sign(Vt[np.arange(k), idx])[:, None]
        self.n_components_, self.n_samples_, self.n_features_in_ = int(k), int(N), int(d)
        self.mean_ = mean.cpu().numpy()
        self.components_ = Vt
        self.singular_values_ = S
        self.explained_variance_ = S ** 2 / max(1, N - 1)
        var = sd.cpu().numpy() ** 2 * (N / max(1, N - 1))     # per-feature variance, ddof = 1
        self.explained_variance_ratio_ = self.explained_variance_ / var.sum()
        comp = torch.from_numpy(np.ascontiguousarray(Vt.astype(np.float32))).to(dev)
        return mm_nt(Xc, comp)                                 # scores = Xc V = U S, [N, k] fp32

    def fit_transform(self, X):
        return self.fit_transform_device(X).cpu().numpy().astype(np.float64)

    def fit(self, X):
        self.fit_transform_device(X)
        return self

    # ---- sklearn's transform pair: on the device for many rows, numpy otherwise ----
    def _on_device(self, n_rows):
        return n_rows >= 2048 and torch.cuda.is_available()

    def transform(self, X):
        X = np.asarray(X) if not torch.is_tensor(X) else X
        if not self._on_device(X.shape[0]):
            return (np.asarray(X, dtype=np.float64) - self.mean_) @ self.components_.T
        dev = torch.device(self.device)
        Xc, _, _ = _center(_as_device_matrix(X, dev), torch.from_numpy(self.mean_).to(dev))
        comp = torch.from_numpy(np.ascontiguousarray(self.components_.astype(np.float32))).to(dev)
        return mm_nt(Xc, comp).cpu().numpy().astype(np.float64)

    def inverse_transform(self, Z):
        Z = np.asarray(Z) if not torch.is_tensor(Z) else Z
        if not self._on_device(Z.shape[0]):
            return np.asarray(Z, dtype=np.float64) @ self.components_ + self.mean_
        dev = torch.device(self.device)
        Zd = torch.as_tensor(np.ascontiguousarray(Z)).to(dev, torch.float32).contiguous()
        comp = torch.from_numpy(np.ascontiguousarray(self.components_.astype(np.float32))).to(dev)
        bias = torch.from_numpy(self.mean_.astype(np.float32)).to(dev)
        return mm_nn(Zd, comp, bias=bias).cpu().numpy().astype(np.float64)


def global_standardise(scores):
    """`preclass(sample, pca=pca)` (axis=None, reference utilities.py:654-678) on the device: ONE mean and ONE population
    standard deviation over all entries of the score matrix; returns the standardised fp32 matrix and (mean, std)."""
    N, k = scores.shape
    R = int(max(1, min(256, (N + 2047) // 2048)))
    part = torch.empty(R * k, dtype=torch.float64, device=scores.device)
    cm = torch.empty(k, dtype=torch.float64, device=scores.device)
    cs = torch.empty(k, dtype=torch.float64, device=scores.device)
    nv._call('jamie_col_stats', nv.ptr(scores), 0, N, k, k, nv.ptr(part), R, nv.ptr(cm), nv.ptr(cs), nv._stream())
    cmh, csh = cm.cpu().numpy(), cs.cpu().numpy()
    m = float(cmh.mean())
    s = float(np.sqrt(np.mean(csh ** 2 + (cmh - m) ** 2)))       # variance of the pooled entries (equal column sizes)
    mean = torch.full((k,), m, dtype=torch.float64, device=scores.device)
    sd = torch.full((k,), s, dtype=torch.float64, device=scores.device)
    out = torch.empty(N, k, dtype=torch.float32, device=scores.device)
    nv._call('jamie_standardise', nv.ptr(scores), 0, N, k, k, nv.ptr(mean), nv.ptr(sd), nv.ptr(out), nv._stream())
    return out, m, s
