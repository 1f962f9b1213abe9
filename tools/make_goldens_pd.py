#!/usr/bin/env python
"""Golden fixtures for the correspondence stage: the REFERENCE's own
`JAMIE.Prime_Dual` (jamie/jamie.py:314-414) and its stage-A/B wiring (`compute_distances` with a sklearn metric ->
`match`, jamie.py:155-177, 224-249, 839-890), imported from /root/reference with the stubs of tools/ref_stubs.py.

Run in the build container only:  python tools/make_goldens_pd.py
Fixtures are data (inputs + the reference's outputs); no reference source is copied."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
import matplotlib  # noqa: E402

matplotlib.use('Agg')
import jamie as ref  # noqa: E402
from sklearn.metrics import pairwise_distances  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def dist_pair(rng, m, n, dx, dy, k=4):
    Zx = rng.standard_normal((m, k))
    X = Zx @ rng.standard_normal((k, dx)) + 0.1 * rng.standard_normal((m, dx))
    Y = Zx[:n] @ rng.standard_normal((k, dy)) + 0.1 * rng.standard_normal((n, dy))
    return X, Y, pairwise_distances(X, metric='euclidean'), pairwise_distances(Y, metric='euclidean')


def pd_case(name, m, n, dx, dy, seed, **kw):
    rng = np.random.default_rng(seed)
    X, Y, Kx, Ky = dist_pair(rng, m, n, dx, dy)
    with contextlib.redirect_stdout(io.StringIO()):
        jm = ref.JAMIE(**kw)
        F = jm.Prime_Dual([Kx, Ky], dx=dx, dy=dy, verbose=False)
    meta = dict(m=m, n=n, dx=dx, dy=dy, epoch_pd=jm.epoch_pd, rho=jm.rho, epsilon=jm.epsilon, delay=jm.delay)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), Kx=Kx, Ky=Ky, F=np.asarray(F, np.float32),
                        meta=np.array(repr(meta)))
    print(name, meta, 'F range', float(F.min()), float(F.max()), 'rowsum mean', float(F.sum(1).mean()))


def pipeline_case(name, m, dx, dy, seed, **kw):
    """Stages A + B through the reference's fit_transform wiring: euclidean distances -> Prime_Dual -> F."""
    rng = np.random.default_rng(seed)
    X, Y, _, _ = dist_pair(rng, m, m, dx, dy)
    rec = {}
    with contextlib.redirect_stdout(io.StringIO()):
        jm = ref.JAMIE(distance_mode='euclidean', **kw)
        jm.dataset = [X, Y]
        jm.dataset_num = 2
        jm.row, jm.col = [m, m], [dx, dy]
        jm.compute_distances(save_dist=True)
        rec['dist'] = [np.asarray(d) for d in jm.dist]
        F = jm.match()[0]
    meta = dict(m=m, dx=dx, dy=dy, epoch_pd=jm.epoch_pd, rho=jm.rho, epsilon=jm.epsilon, delay=jm.delay)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), X=X, Y=Y, dist0=rec['dist'][0], dist1=rec['dist'][1],
                        F=np.asarray(F, np.float32), meta=np.array(repr(meta)))
    print(name, meta, 'F range', float(F.min()), float(F.max()))


if __name__ == '__main__':
    torch.set_num_threads(4)
    pd_case('pd1_delay0', 48, 40, 30, 20, 1, epoch_pd=150)
    pd_case('pd2_delay', 64, 64, 24, 36, 2, epoch_pd=120, delay=40, epsilon=0.01, rho=5)
    pipeline_case('pd3_pipeline', 56, 20, 28, 3, epoch_pd=100)
