"""Stub modules so the (read-only, Python) reference at /root/reference can be imported in the
build container, where `unioncom`, `anndata`, `umap`, `seaborn`, `adjustText`, `brokenaxes` are absent
and there is no network.  Used ONLY by tools/make_goldens.py (fixture generation); nothing here ships
to the GPU box's run-time paths and nothing of the reference's source is copied.

The stubs carry no hot-path arithmetic: `unioncom.UnionCom.UnionCom.__init__` only stores constructor
attributes and `init_random_seed` seeds python `random` + torch, which is all
jamie/jamie.py:142 relies on.
"""
import random
import sys
import types

import torch

REFERENCE_ROOT = '/root/reference'


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if 'unioncom' in sys.modules:
        return
    # anndata._core.anndata.AnnData  (isinstance check, jamie/jamie.py:147)
    ad = _mod('anndata')
    core = _mod('anndata._core')
    adad = _mod('anndata._core.anndata')

    class AnnData:  # never instantiated
        pass
    adad.AnnData = AnnData
    core.anndata = adad
    ad._core = core

    # umap.UMAP (only used when model_pca='umap')
    um = _mod('umap')

    class UMAP:
        def __init__(self, *a, **k):
            raise RuntimeError('umap is not available in this container')
    um.UMAP = UMAP

    # unioncom
    uc = _mod('unioncom')
    ucm = _mod('unioncom.UnionCom')
    ucu = _mod('unioncom.utils')

    class UnionCom:
        # defaults of unioncom==0.4.0 (requirements.txt:185), as recalled in SURVEY.md §5
        def __init__(self, distance_mode='geodesic', project_mode='tsne', integration_type='MultiOmics',
                     epoch_pd=2000, epoch_DNN=100, epsilon=0.001, lr=0.001, batch_size=100, rho=10,
                     log_DNN=10, log_pd=1000, manual_seed=666, delay=0, kmax=40, beta=1,
                     perplexity=30, output_dim=32, **unused):
            self.distance_mode = distance_mode
            self.project_mode = project_mode
            self.integration_type = integration_type
            self.epoch_pd = epoch_pd
            self.epoch_DNN = epoch_DNN
            self.epsilon = epsilon
            self.lr = lr
            self.batch_size = batch_size
            self.rho = rho
            self.log_DNN = log_DNN
            self.log_pd = log_pd
            self.manual_seed = manual_seed
            self.delay = delay
            self.kmax = kmax
            self.beta = beta
            self.perplexity = perplexity
            self.output_dim = output_dim

    def init_random_seed(manual_seed):
        seed = random.randint(1, 10000) if manual_seed is None else manual_seed
        print('use random seed: {}'.format(seed))
        random.seed(seed)
        torch.manual_seed(seed)

    def _absent(*a, **k):
        raise RuntimeError('unioncom stage A/B helpers are out of scope and not stubbed')

    ucm.UnionCom = UnionCom
    ucu.init_random_seed = init_random_seed
    ucu.geodesic_distances = _absent
    ucu.joint_probabilities = _absent
    uc.UnionCom = ucm
    uc.utils = ucu

    # plotting-only deps pulled by jamie/__init__.py:3 -> evaluation.py:4-5,11
    _mod('seaborn')
    at = _mod('adjustText')
    at.adjust_text = _absent
    ba = _mod('brokenaxes')
    ba.brokenaxes = _absent

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
