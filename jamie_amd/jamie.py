"""`JAMIE` facade with the reference's constructor and method surface (reference jamie/jamie.py:29-972)
for the coupled-VAE path: `fit_transform` / `fit`, `transform`, `transform_one`, `modal_predict` /
`impute`, `save_model`, `load_model`, `loss_history`.

Scope: `project_mode='jamie'`.  The correspondence `F` is absent (`use_f_tilde=False`), supplied as `match_result`, or
computed like the reference does: stage A (`compute_distances`, jamie.py:839-890) on the host and stage B
(`match` / `Prime_Dual`, jamie.py:224-414) on the GPU (jamie_amd/correspondence.py).
Unlike the reference no N x N matrix is ever formed when `P` is the identity (the reference's default for
equally sized datasets, jamie.py:425-426): the B x B block `P[idx][:, idx]` is index equality.
"""
import random
import warnings

import numpy as np
import scipy.sparse as sp
import torch

from . import _native as nv
from . import distributed as jd
from .engine import LOSS_NAMES, TrainEngine, kl_anneal
from .model import edModelVar
from .utilities import preclass, time_logger

_DISTANCE_MODES = [
    'euclidean', 'l2', 'l1', 'manhattan', 'cityblock', 'braycurtis', 'canberra', 'chebyshev', 'correlation',
    'cosine', 'dice', 'hamming', 'jaccard', 'kulsinski', 'mahalanobis', 'matching', 'minkowski',
    'rogerstanimoto', 'russellrao', 'seuclidean', 'sokalmichener', 'sokalsneath', 'sqeuclidean', 'yule',
    'wminkowski', 'nan_euclidean', 'haversine', 'geodesic', 'spearman', 'pearson']


def init_random_seed(manual_seed):
    """unioncom.utils.init_random_seed (called at reference jamie.py:142): seeds python `random` and torch."""
    seed = random.randint(1, 10000) if manual_seed is None else manual_seed
    print('use random seed: {}'.format(seed))
    random.seed(seed)
    torch.manual_seed(seed)
    return seed


class JAMIE:
    """MI355X drop-in for `jamie.JAMIE` (coupled-VAE path).

    Constructor keywords are the reference's (jamie.py:38-63) plus the UnionCom keywords it forwards
    (jamie.py:99-111; defaults from unioncom==0.4.0).  Extra, MI355X-only keywords:
      sampler      'auto' (default): 'device' -- Philox batch sampler on the GPU, the step replayed from a recorded launch
                   plan, no host work per step -- unless an explicit-noise seam is installed (`_noise_source`, the fixture
                   replays), then 'numpy'; 'numpy': the reference's `np.random.choice` index stream under `np.random.seed`
                   (drawn on the host every step: ~1 ms at 100k cells); 'device'.  Dropout masks and the reparameterisation noise
                   come from device Philox streams either way, so a production run does not replay the reference's numbers
                   whichever sampler draws the rows: the numpy stream matters to the fixture replays only
      distributed  True -> one process per GPU under torchrun; cells are sharded by rows and the flat
                   gradient is all-reduced once per step over RCCL
      compute_dtype 'f32' (default: exact-fp32 MFMA, the parity configuration) or 'bf16' (bf16 MFMA GEMMs with
                   fp32 accumulation, master weights, optimiser, BatchNorm and losses; feature counts, latent
                   size and batch size must be multiples of 8)
      dp_optimizer 'replicated' (default), 'sharded' or 'auto' (distributed runs): replicated = ONE all-reduce of the flat
                   gradient per step and the full update on every rank (the arrangement north_star names; the only one that
                   has run on more than one GPU over RCCL so far); sharded = the large gradient regions are
                   reduce-scattered, every rank runs clip + Adam over 1/world of the large weight matrices and the updated
                   weights are all-gathered under the next forward pass (distributed.ShardedGradExchange); auto = sharded
                   where it applies (batch_step=True, a world size that divides the regions, no transposed weight copies),
                   else replicated.  With the sharded optimiser `save_checkpoint` / evaluation gather the packed state with
                   collectives: every rank must call them
      grad_comm_dtype 'auto' (default: the compute dtype), 'f32' or 'bf16': precision of the gradient all-reduce
                   messages when distributed (bf16 halves the 4 P bytes exchanged per step)
    """

    def __init__(self, match_result=None, PF_Ratio=None, corr_method='unioncom', dist_method='euclidean',
                 in_place=False, loss_weights=None, model_pca='pca', model_class=edModelVar, model_lr=1e-3,
                 dropout=None, pca_dim=2 * [512], batch_step=True, use_f_tilde=True, use_early_stop=True,
                 min_epochs=2500, min_increment=1e-8, max_steps_without_increment=500, debug=False,
                 log_debug=100, record_loss=True, enable_memory_logging=False, device='cuda',
                 sampler='auto', distributed=False, compute_dtype='f32', grad_comm_dtype='auto', dp_optimizer='replicated',
                 preprocess='host', checkpoint_path=None, checkpoint_every=0, **kwargs):
        self.match_result = match_result
        self.PF_Ratio = PF_Ratio
        self.corr_method = corr_method
        self.dist_method = dist_method
        self.in_place = in_place
        self.loss_weights = loss_weights
        self.model_pca = model_pca
        self.model_class = model_class
        self.model_lr = model_lr
        self.dropout = dropout
        self.pca_dim = pca_dim
        self.batch_step = batch_step
        self.use_f_tilde = use_f_tilde
        self.use_early_stop = use_early_stop
        self.min_epochs = min_epochs
        self.min_increment = min_increment
        self.max_steps_without_increment = max_steps_without_increment
        self.debug = debug
        self.log_debug = log_debug
        self.record_loss = record_loss
        self.enable_memory_logging = enable_memory_logging
        if device == 'cpu':
            raise nv.JamieHipError("jamie_amd runs on the MI355X only; use device='cuda' (no CPU fallback)")
        self.device = device
        if sampler not in ('auto', 'numpy', 'device'):
            raise ValueError("sampler must be 'auto', 'numpy' or 'device'")
        self.sampler = sampler
        self.distributed = distributed
        self.compute_dtype = compute_dtype
        if preprocess not in ('host', 'device'):
            raise ValueError("preprocess must be 'host' (numpy fp64, the reference's arithmetic) or 'device'")
        self.preprocess = preprocess
        # training checkpoints (the reference only pickles the finished model, jamie.py:967-972):
        # every `checkpoint_every` epochs the full training state goes to `checkpoint_path`;
        # fit_transform(..., resume_from=path) continues from it and ends bit-identical to an uninterrupted run
        self.checkpoint_path, self.checkpoint_every = checkpoint_path, int(checkpoint_every or 0)
        self._resume_from = None
        if grad_comm_dtype not in ('auto', 'f32', 'bf16'):
            raise ValueError("grad_comm_dtype must be 'auto', 'f32' or 'bf16'")
        if dp_optimizer not in ('auto', 'sharded', 'replicated'):
            raise ValueError("dp_optimizer must be 'auto', 'sharded' or 'replicated'")
        self.dp_optimizer = dp_optimizer
        self.grad_comm_dtype = compute_dtype if grad_comm_dtype == 'auto' else grad_comm_dtype
        # UnionCom attributes (reference jamie.py:99-111 defaults, then unioncom 0.4.0's)
        defaults = {'project_mode': 'jamie', 'log_pd': 500, 'lr': 1e-3, 'epoch_DNN': 10000, 'log_DNN': 500,
                    'batch_size': 512, 'epoch_pd': 2000, 'epsilon': 1e-3, 'rho': 10, 'beta': 1, 'perplexity': 30,
                    'manual_seed': 666, 'delay': 0, 'kmax': 40, 'output_dim': 32, 'distance_mode': 'geodesic',
                    'integration_type': 'MultiOmics'}
        for k, v in defaults.items():
            setattr(self, k, kwargs.pop(k, v))
        if kwargs:
            raise TypeError(f'unexpected keyword arguments: {sorted(kwargs)}')
        self.model = None
        self.engine = None
        self.loss_history = {}
        # test seam: callable(step_number) -> {'eps', 'enc_masks', 'dec_masks'} device tensors fed to the step instead of
        # the Philox streams, so that the loop users call can be replayed against the reference's fixtures
        self._noise_source = None

    # ------------------------------------------------------------------------------------------
    def fit_transform(self, dataset=None, P=None, resume_from=None):
        """reference jamie.py:113-222.  `resume_from`: a checkpoint written by this class (`checkpoint_path`)."""
        self.P = P
        self._resume_from = resume_from
        if self.integration_type not in ['MultiOmics']:
            raise Exception('integration_type error! Enter MultiOmics.')
        if self.distance_mode not in _DISTANCE_MODES:
            raise Exception('distance_mode error! Enter a correct distance_mode.')
        if self.project_mode not in ('jamie', 'tsne'):
            raise Exception("Choose correct project_mode: 'nlma', 'tsne'.")
        assert self.model_pca in ('pca', 'umap')
        if self.project_mode == 'tsne':
            raise NotImplementedError("project_mode='tsne' is outside the accelerated path (SURVEY.md §8)")
        time = time_logger(memory_usage=self.enable_memory_logging, sync=torch.cuda.synchronize)   # jamie.py:141
        init_random_seed(self.manual_seed)                         # jamie.py:142
        self.dataset = dataset
        self.dataset_annotation = None
        if hasattr(self.dataset[0], 'X') and not isinstance(self.dataset[0], np.ndarray):   # AnnData
            self.dataset = [d.X for d in self.dataset]
            self.dataset_annotation = dataset
        if not self.in_place:
            self.dataset = self.dataset * 1                        # shallow copy, jamie.py:152-153
        self.dataset_num = len(self.dataset)
        self.row = [np.shape(d)[0] for d in self.dataset]
        self.col = [np.shape(d)[1] for d in self.dataset]
        # stage A (host, as in the reference) and stage B (Prime_Dual on the GPU): jamie.py:162-177
        if self.match_result is None and self.use_f_tilde:
            if self.dataset_num != 2:
                raise NotImplementedError('use_f_tilde=True follows the reference: two modalities')
            self.compute_distances(save_dist=True)
        time.log('Distance')
        if self.match_result is None and self.use_f_tilde:
            self.match_result = self.match()
        time.log('Correspondence')
        integrated = self.project_jamie()
        time.log('Mapping')
        print('-' * 33)
        print('JAMIE Done!')
        time.aggregate()
        print()
        return integrated

    # ---- stages A / B (reference jamie.py:224-249, 314-414, 839-890) ----
    def compute_distances(self, save_dist=True):
        """Cell x cell distance matrix of every modality (host numpy / scipy / sklearn, as in the reference)."""
        from .utilities import distance_matrix
        if save_dist:
            self.dist = []
        print('Shape of Raw data')
        for i in range(self.dataset_num):
            print('Dataset {}:'.format(i), np.shape(self.dataset[i]))
        self.distance_function = lambda df: distance_matrix(df, self.distance_mode, self.kmax)   # noqa: E731
        if save_dist:
            self.dist = [self.distance_function(d) for d in self.dataset]

    def Prime_Dual(self, dist, dx=None, dy=None, verbose=True):
        """reference jamie.py:314-414, on the MI355X (jamie_amd/correspondence.py); returns F as numpy float32."""
        from .correspondence import prime_dual
        if self.integration_type != 'MultiOmics':
            raise NotImplementedError("Prime_Dual: integration_type 'MultiOmics' (the only one fit_transform accepts)")
        return prime_dual(dist, dx, dy, epoch_pd=self.epoch_pd, rho=self.rho, epsilon=self.epsilon, delay=self.delay,
                          log_pd=self.log_pd, verbose=verbose, device=self.device)

    def match(self):
        """reference jamie.py:224-249."""
        print('Device:', self.device)
        cor_pairs = []
        for i in range(self.dataset_num):
            for j in range(i + 1, self.dataset_num):
                print('-' * 33)
                print(f'Find correspondence between Dataset {i + 1} and Dataset {j + 1}')
                if self.corr_method != 'unioncom':
                    raise NotImplementedError("corr_method='jamie' is a work in progress in the reference "
                                              '(jamie.py:241-246) and is not built')
                cor_pairs.append(self.Prime_Dual([self.dist[i], self.dist[j]], dx=self.col[i], dy=self.col[j]))
        print('Finished Matching!')
        return cor_pairs

    def fit(self, dataset=None, P=None, resume_from=None):
        """north_star spelling: train, return self."""
        self._last_embedding = self.fit_transform(dataset, P, resume_from=resume_from)
        return self

    # ------------------------------------------------------------------------------------------
    def _build_preprocessing(self):
        """reference jamie.py:434-469."""
        pre = []
        if self.pca_dim is not None:
            for dim, data in zip(self.pca_dim, self.dataset):
                if dim is not None:
                    from sklearn.decomposition import PCA
                    if min(*data.shape) < dim:
                        warnings.warn(f'PCA dim must be lower than {min(*data.shape)}, found {dim}, '
                                      f'adjusting to compensate.')
                        dim = min(*data.shape)
                    if self.model_pca != 'pca':
                        raise NotImplementedError("model_pca='umap' needs umap-learn (absent)")
                    pca = PCA(n_components=dim)
                    sample = pca.fit_transform(data)
                    pre.append(preclass(sample, pca=pca))
                else:
                    pre.append(preclass(np.asarray(data), axis=0))
        else:
            pre = [preclass(np.asarray(d), axis=0) for d in self.dataset]
        return pre

    def project_jamie(self, W=None):
        """The training loop, reference jamie.py:416-804."""
        print('-' * 33)
        print('Train coupled autoencoders')
        # the reference asserts two modalities (jamie.py:420); 3-4 fully paired ones run the build-defined
        # generalisation (identity correspondence, F = 0)
        assert 2 <= self.dataset_num <= 4, 'Currently only compatible with 2 (reference) to 4 modalities.'
        if self.dataset_num > 2 and (self.P is not None or self.use_f_tilde or len(set(self.row)) != 1):
            raise NotImplementedError('more than two modalities need fully paired cells (P=None, use_f_tilde=False)')
        dev = torch.device(self.device)
        rank, world = 0, 1
        allreduce = None
        if self.distributed:
            rank, world, _ = jd.init_from_env()
            allreduce = jd.OverlappedGradAllReduce(comm_dtype=torch.bfloat16 if self.grad_comm_dtype == 'bf16' else None)
        # ---- P / F (never densified for the identity) ----
        P_dense = P_csr = None
        if self.P is None:
            method = 'diag' if self.row[0] == self.row[1] else 'zeros'      # jamie.py:423-428, 518-533
        elif sp.issparse(self.P):
            # sparse partial correspondence: CSR on the device, the B x B block of a batch is
            # looked up by jamie_csr_block -- no N x N array at any point
            Pc = sp.csr_matrix(self.P, dtype=np.float32)
            Pc.sum_duplicates()
            Pc.sort_indices()
            if Pc.shape[0] == Pc.shape[1] and Pc.nnz == Pc.shape[0] and bool((Pc.diagonal() == 1).all()):
                method = 'diag'
            elif Pc.nnz and float(np.abs(Pc.data).sum()) != 0:
                method = 'hybrid'                                             # corrected sampler, see the dense branch
                coo = Pc.tocoo()
                keep = coo.data > 0
                self.corr_samples = np.stack([coo.row[keep], coo.col[keep]], axis=1)
                self.num_corr = len(self.corr_samples)
                self.true_ratio = .8                                          # jamie.py:529
                P_csr = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
                              (Pc.indptr.astype(np.int32), Pc.indices.astype(np.int32), Pc.data.astype(np.float32)))
            else:
                method = 'zeros'
        else:
            P_dense = torch.as_tensor(np.asarray(self.P), dtype=torch.float32, device=dev)
            if P_dense.shape[0] == P_dense.shape[1] and torch.abs(
                    P_dense - torch.eye(self.row[0], device=dev)).sum() == 0:
                method, P_dense = 'diag', None
            elif torch.abs(P_dense).sum() != 0:
                # partial correspondence (reference jamie.py:523-530), CORRECTED: the reference sets
                # `num_corr = len(corr_samples[0])` (== 2, the pair width) and indexes `corr_samples[i]` (the i-th
                # pair) per modality, so it never samples the known pairs as intended; here num_corr is the number
                # of known pairs and column i of the pair list feeds modality i.
                method = 'hybrid'
                self.corr_samples = np.argwhere(np.asarray(self.P) > 0)
                self.num_corr = len(self.corr_samples)
                self.true_ratio = .8                                          # jamie.py:529
            else:
                method = 'zeros'
        self.sampling_method = method
        F_dense = None
        if self.use_f_tilde:
            mr = self.match_result[0]
            F_dense = (mr.to(dev, torch.float32) if torch.is_tensor(mr)
                       else torch.as_tensor(np.asarray(mr), dtype=torch.float32, device=dev))
        timer = time_logger(memory_usage=self.enable_memory_logging, sync=torch.cuda.synchronize)    # jamie.py:433
        # per-step labels of the reference ('Get subset samples', 'Step'; jamie.py:601, 742): a synchronising timer inside
        # the loop would block the host twice per step, so it only runs when `debug` asks for the per-phase table
        step_timer = timer if self.debug else time_logger(record=False)
        # ---- preprocessing (host, numpy/sklearn like the reference) ----
        dev_pre = self.preprocess == 'device'
        if dev_pre:
            # on the GPU: per-feature standardisation (jamie_col_stats / jamie_standardise: fp64
            # statistics in numpy's two-pass order) or, where `pca_dim` names a dimension (the reference's default,
            # jamie.py:50, 436-457), randomized PCA on the fp32 MFMA GEMM (jamie_amd/pca.py) followed by the global
            # scaling of `preclass(sample, pca=pca)`; fp32 cells are written straight into the resident training matrices.
            # The host keeps the `preclass` objects (statistics, fitted PCA) for transform / inverse_transform of new data
            from .pca import DevicePCA, global_standardise
            pre, data_dev = [], []
            dims_pca = self.pca_dim if self.pca_dim is not None else [None] * len(self.dataset)
            for dim, x in zip(dims_pca, self.dataset):
                xa = np.ascontiguousarray(np.asarray(x))
                if xa.dtype not in (np.float32, np.float64):
                    xa = xa.astype(np.float64)
                if dim is None:
                    out, mean, sd = nv.standardise_columns(torch.from_numpy(xa).to(dev))
                    pre.append(preclass.from_stats(mean.cpu().numpy(), sd.cpu().numpy(), axis=0))
                else:
                    if self.model_pca != 'pca':
                        raise NotImplementedError("model_pca='umap' needs umap-learn (absent)")
                    if min(*xa.shape) < dim:
                        warnings.warn(f'PCA dim must be lower than {min(*xa.shape)}, found {dim}, '
                                      f'adjusting to compensate.')
                        dim = min(*xa.shape)
                    pca = DevicePCA(dim, device=self.device)        # random_state=None: numpy's global RNG, like sklearn's
                    out, m, sdev = global_standardise(pca.fit_transform_device(torch.from_numpy(xa).to(dev)))
                    pre.append(preclass.from_stats(m, sdev, axis=None, pca=pca))
                data_dev.append(out)
            self.dataset = [d.cpu().numpy() for d in data_dev]                # the reference keeps the transformed cells
        else:
            pre = self._build_preprocessing()
            self.dataset = [p.transform(np.asarray(x)) for p, x in zip(pre, self.dataset)]
        self.col = [x.shape[1] for x in self.dataset]
        # ---- model / engine ----
        extra = {}
        if self.compute_dtype == 'bf16' and self.model_class is edModelVar and any(c % 8 for c in self.col):
            extra['pad_features'] = 8        # bf16 GEMM operands need feature counts that are multiples of 8 (model.py)
        self.model = self.model_class(self.col, self.output_dim, preprocessing=[p.transform for p in pre],
                                      preprocessing_inverse=[p.inverse_transform for p in pre],
                                      dropout=self.dropout, **extra).to(self.device)
        if world > 1:
            jd.broadcast_flat(self.model.flat)
        self.model.train()
        data_all = data_dev if dev_pre else [torch.from_numpy(np.ascontiguousarray(x)).float() for x in self.dataset]
        # row shards for data parallelism ('diag' keeps the same rows of both modalities on one rank)
        bounds = [jd.shard_bounds(r, rank, world) for r in self.row]
        data = [d[lo:hi].to(dev).contiguous() for d, (lo, hi) in zip(data_all, bounds)]
        rows = [hi - lo for lo, hi in bounds]
        # every rank must run the same number of steps per epoch (each step is a collective) with the same batch size:
        # both follow from the SMALLEST shard (shard sizes differ by at most one row), not from this rank's own
        rows_all = [[b[1] - b[0] for b in (jd.shard_bounds(r, k, world) for r in self.row)] for k in range(world)]
        len_dataloader = min(int(np.max(rk) / self.batch_size) for rk in rows_all)   # jamie.py:511-514
        if len_dataloader == 0:
            len_dataloader = 1
            self.batch_size = min(int(np.max(rk)) for rk in rows_all)
        B = int(self.batch_size)
        if method == 'hybrid' and world > 1:
            # data parallel partial correspondence: a rank samples the known pairs whose two cells both live in its row
            # shards (contiguous shards of every modality; a pair that straddles two ranks is only reachable through the
            # unpaired draws of its two cells); indices become shard-local
            cs = self.corr_samples
            keep = ((cs[:, 0] >= bounds[0][0]) & (cs[:, 0] < bounds[0][1]) & (cs[:, 1] >= bounds[1][0]) & (cs[:, 1] < bounds[1][1]))
            self.corr_samples = cs[keep] - np.array([bounds[0][0], bounds[1][0]])
            self.num_corr = len(self.corr_samples)
            if self.num_corr == 0:
                raise ValueError(f'rank {rank}: no known pair lies inside its row shards; shard paired cells together')
        self.PF_Ratio = 1 if self.PF_Ratio is None else self.PF_Ratio        # jamie.py:517
        eng = TrainEngine(self.model, B, lr=self.model_lr, loss_weights=self.loss_weights,
                          dist_method=self.dist_method, seed=int(self.manual_seed) + 7919 * rank,
                          world_size=world, compute_dtype=self.compute_dtype)
        eng.accumulate = False
        if not self.batch_step:
            eng.set_grad_bf16(False)        # gradients accumulate over the batches of an epoch (jamie.py:734-749): fp32 buffer
        self.engine = eng
        data = eng.pad_cells(data)
        rep = min(self.col) < B and self.dataset_num == 2                    # jamie.py:553 (sic); M > 2: never
        need_block = ((method == 'diag' and rep) or method == 'zeros' or F_dense is not None or P_dense is not None
                      or P_csr is not None or self.PF_Ratio != 1)
        if self.dataset_num > 2 and need_block:
            raise NotImplementedError('more than two modalities: min(features) must be >= batch_size (no duplicates)')
        idx_dev = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(self.dataset_num)]
        blk_F = torch.empty(B, B, device=dev) if F_dense is not None else None
        blk_mix = torch.empty(B, B, device=dev) if self.PF_Ratio != 1 else None
        best_running_loss = np.inf
        streak = 0
        if self.record_loss:
            self.loss_history = {}
        # fast path: device sampler, 'diag' sampling, no dense P/F blocks -> the step is a fixed launch sequence on
        # static buffers: record it once and replay it (one foreign call per launch, nothing rebuilt per step)
        plan = None
        sampler = self.sampler if self.sampler != 'auto' else ('numpy' if self._noise_source is not None else 'device')
        use_plan = (sampler == 'device' and method == 'diag' and P_dense is None and F_dense is None
                    and P_csr is None and self.PF_Ratio == 1 and self.batch_step)
        # partial correspondence from a sparse P with the device sampler: pair / rest candidates from jamie_sample_indices,
        # jamie_hybrid_assemble picks per slot, jamie_csr_block builds the [B,B] block -- the whole step stays on the GPU and
        # is recorded as a plan too (the numpy sampler costs 2 ms of host time per step at 100k cells: np.random.choice without
        # replacement permutes all N rows)
        plan_hybrid = (sampler == 'device' and method == 'hybrid' and P_csr is not None and F_dense is None
                       and self.PF_Ratio == 1 and self.batch_step and world == 1 and B <= 2048)
        if plan_hybrid:
            pairs_dev = torch.from_numpy(np.ascontiguousarray(self.corr_samples.astype(np.int32))).to(dev)
            hy = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(3)]

            def hybrid_step():
                nv.sample_indices_group([nv.sample_args(hy[0], self.num_corr, 0, rep or self.num_corr < B, 202),
                                         nv.sample_args(hy[1], rows[0], 0, rep, 200),
                                         nv.sample_args(hy[2], rows[1], 0, rep, 201)], eng.state)
                nv.hybrid_assemble(pairs_dev, hy[0], hy[1], hy[2], self.num_corr, self.true_ratio, eng.state, 203,
                                   idx_dev[0], idx_dev[1])
                eng.load_batch(data, idx_dev)
                nv.csr_block(*P_csr, idx_dev[0], idx_dev[1], eng.corr, bounds[0][0], bounds[1][0])
                eng.step(eng.corr, None, None, allreduce)
        epoch_sum = torch.zeros((), device=dev)      # batch_step=False: epoch loss = mean of the batch losses (jamie.py:728)
        start_epoch = 0
        if self._resume_from is not None:
            start_epoch, best_running_loss, streak = self._load_checkpoint(self._resume_from, eng)
            self._resume_from = None
        if allreduce is not None and world > 1 and self.dp_optimizer != 'replicated' and self.batch_step:
            # (after a resume: the packed pieces are cut from the restored parameters and moments)
            try:
                ex = jd.ShardedGradExchange(comm_dtype=allreduce.comm_dtype)
                eng.enable_sharded_optimizer(ex)
                allreduce = ex
            except ValueError:
                if self.dp_optimizer == 'sharded':
                    raise
        timer.log('Setup')
        n_steps = start_epoch * len_dataloader
        if self._noise_source is not None and (use_plan or plan_hybrid):
            raise ValueError('explicit noise needs the eager path (sampler="numpy")')
        for epoch in range(start_epoch, self.epoch_DNN):                      # jamie.py:546
            eng.set_kl_anneal(kl_anneal(epoch, self.min_epochs, self.epoch_DNN))
            eng.reset_best()
            for batch_idx in range(len_dataloader):
                if use_plan:
                    if plan is None:
                        plan = eng.make_plan(data, idx_dev[0], rows[0], rep, allreduce)
                    else:
                        eng.run_plan(plan)
                    continue
                if plan_hybrid:
                    if plan is None:
                        nv.begin_record()
                        try:
                            hybrid_step()                  # the recording step is a real step
                        finally:
                            plan = nv.end_record()
                    else:
                        eng.run_plan(plan)
                    continue
                # ---- sampler (jamie.py:552-583) ----
                if method == 'hybrid':                                         # jamie.py:559-573 (corrected)
                    corr_sample_num = int(min(np.sum(np.random.rand(B) < self.true_ratio), self.num_corr))
                    pairs = self.corr_samples[np.random.choice(self.num_corr, corr_sample_num, replace=rep)]
                    for i in range(2):
                        rest = np.random.choice(rows[i], B - corr_sample_num, replace=rep)
                        s = np.concatenate([pairs[:, i], rest], axis=0)
                        idx_dev[i].copy_(torch.from_numpy(s.astype(np.int32)))
                elif sampler == 'numpy':
                    if method == 'diag':
                        s = np.random.choice(rows[0], B, replace=rep)      # (= choice(range(N), ...): the same stream
                                                                           #  without the 100k-element list conversion)
                        idx_dev[0].copy_(torch.from_numpy(s.astype(np.int32)))
                        for j in range(1, self.dataset_num):
                            idx_dev[j].copy_(idx_dev[0])
                    else:
                        for i in range(2):
                            s = np.random.choice(rows[i], B, replace=rep)
                            idx_dev[i].copy_(torch.from_numpy(s.astype(np.int32)))
                else:
                    nv.sample_indices(idx_dev[0], rows[0], 0, rep, eng.state, 200)
                    if method == 'diag':
                        for j in range(1, self.dataset_num):
                            idx_dev[j].copy_(idx_dev[0])
                    else:
                        nv.sample_indices(idx_dev[1], rows[1], 0, rep, eng.state, 201)
                eng.load_batch(data, idx_dev)
                # ---- P / F blocks (jamie.py:585-604) ----
                corr = Fblk = None
                if need_block:
                    if P_csr is not None:
                        nv.csr_block(*P_csr, idx_dev[0], idx_dev[1], eng.corr, bounds[0][0], bounds[1][0])
                    elif P_dense is not None:         # one B x B gather + row normalisation (no [B, N] slab)
                        nv.dense_block(P_dense, idx_dev[0], idx_dev[1], eng.corr, bounds[0][0], bounds[1][0])
                    elif method == 'diag':
                        nv.corr_from_indices(idx_dev[0], idx_dev[1], eng.corr)
                    else:
                        eng.corr.zero_()
                    corr = eng.corr
                    if F_dense is not None:
                        nv.dense_block(F_dense, idx_dev[0], idx_dev[1], blk_F, bounds[0][0], bounds[1][0])
                        Fblk = blk_F
                    if self.PF_Ratio != 1:                                     # jamie.py:604
                        nv.axpby(blk_mix, self.PF_Ratio, eng.corr, 1 - self.PF_Ratio, Fblk)
                        corr = blk_mix
                step_timer.log('Get subset samples')
                noise = None if self._noise_source is None else self._noise_source(n_steps)
                n_steps += 1
                if self.batch_step:
                    eng.step(corr, Fblk, noise, allreduce)
                else:
                    # jamie.py:734-749: every batch back-propagates into the same gradient buffers, ONE clip + Adam
                    # step per epoch; the gradient all-reduce rides on the last batch's backward
                    last = batch_idx == len_dataloader - 1
                    eng.accumulate = batch_idx > 0
                    eng.forward_backward(corr, Fblk, noise, allreduce if last else None)
                    epoch_sum += eng.losses[4]
                    if not last:
                        eng.state[0] += 1          # the Philox step only advances with the optimiser: fresh noise per batch
                    else:
                        eng.state[0] -= len_dataloader - 1
                    if last:
                        if allreduce is not None:
                            allreduce.finish() if hasattr(allreduce, 'finish') else allreduce(eng.grad)
                        eng.accumulate = False
                        eng.optimizer_step()
                step_timer.log('Step')
            # ---- per-epoch bookkeeping (one device read per epoch; jamie.py:751-792) ----
            ls, total, best_batch_loss = eng.read_losses()
            if not self.batch_step:                    # jamie.py:778-781: the early-stop criterion is the epoch loss
                best_batch_loss = total = float(epoch_sum) / len_dataloader
                epoch_sum.zero_()
            if self.record_loss:
                for name, lo in zip(LOSS_NAMES, ls):
                    self.loss_history.setdefault(name, []).append(lo)
            if (epoch + 1) % self.log_debug == 0 and self.debug:
                print('  '.join(f'{n}: {lo:.4f}' for n, lo in zip(LOSS_NAMES, ls)))
            if (epoch + 1) % self.log_DNN == 0:
                print(f'epoch:[{epoch + 1:d}/{self.epoch_DNN}]: loss:{total:4f}')
            if epoch > self.min_epochs:
                if world > 1:
                    # the ranks train on different shards, so their losses differ; the stop decision must not (a rank that
                    # left the loop would leave the others waiting in the gradient all-reduce): mean over ranks, one scalar
                    # per epoch, outside the step
                    best_batch_loss = jd.mean_scalar(best_batch_loss, dev)
                if best_running_loss - best_batch_loss > self.min_increment:
                    best_running_loss = best_batch_loss
                    streak = 0
                else:
                    streak += 1
                if streak >= self.max_steps_without_increment and self.use_early_stop:
                    break
            if self.checkpoint_every and self.checkpoint_path and (epoch + 1) % self.checkpoint_every == 0:
                eng.flush(collective=True)       # (every rank: a sharded optimiser gathers its pieces with collectives)
                if world > 1:
                    jd.average_(self.model.bn_flat)
                if rank == 0:
                    self.save_checkpoint(self.checkpoint_path, epoch + 1, best_running_loss, streak)
        eng.flush(collective=True)
        if world > 1:
            # BatchNorm running statistics are per-rank during training (no SyncBN: one collective per step); the model
            # that is evaluated / saved carries their average, so every rank returns the same embeddings
            jd.average_(self.model.bn_flat)
        self.model.eval()
        out = [self.model.embed(data_all[i], i).cpu().numpy() for i in range(self.dataset_num)]   # jamie.py:794-799
        timer.log('Output')
        print('Finished Mapping!')
        if self.debug:
            timer.aggregate()
        return out

    # ------------------------------------------------------------------------------------------
    def modal_predict(self, data, modality, pre_transformed=False):
        """reference jamie.py:806-815 (returns float64: the inverse scaling promotes)."""
        assert self.model is not None, 'Model must be trained before modal prediction.'
        to_modality = (modality + 1) % self.dataset_num
        if not pre_transformed:
            data = self.model.preprocessing[modality](np.asarray(data))
        self.model.eval()
        decoded = self.model.impute(torch.as_tensor(np.asarray(data)).float(), compose=[modality, to_modality])
        return np.array(self.model.preprocessing_inverse[to_modality](decoded.detach().cpu()))

    def impute(self, data, modality, pre_transformed=False):
        """north_star spelling of `modal_predict`."""
        return self.modal_predict(data, modality, pre_transformed)

    def transform(self, dataset, corr=None, pre_transformed=False):
        """reference jamie.py:817-829.  In eval mode the returned embeddings are the `mu`s, so `corr` has no
        effect ("Doesn't actually do anything", jamie.py:820) and no N x N matrix is built."""
        if not pre_transformed:
            dataset = [self.model.preprocessing[i](np.asarray(dataset[i])) for i in range(len(dataset))]
        self.model.eval()
        return [self.model.embed(torch.as_tensor(np.asarray(d)).float(), i).cpu().numpy()
                for i, d in enumerate(dataset)]

    def transform_one(self, data, i, pre_transformed=False):
        """reference jamie.py:831-837."""
        if not pre_transformed:
            data = self.model.preprocessing[i](np.asarray(data))
        self.model.eval()
        return self.model.embed(torch.as_tensor(np.asarray(data)).float(), i).cpu().numpy()

    # ------------------------------------------------------------------------------------------
    def save_model(self, f):
        """reference jamie.py:967-968.  Saved as state_dict + preprocessing objects (the reference's
        whole-module pickle no longer loads on torch >= 2.6)."""
        m = self.model
        torch.save({'format': 'jamie_amd.v1', 'input_dim': m.input_dim, 'output_dim': m.output_dim,
                    'dropout': m.dropout, 'state_dict': {k: v.cpu() for k, v in m.state_dict().items()},
                    'preprocessing': m.preprocessing, 'preprocessing_inverse': m.preprocessing_inverse}, f)

    def save_checkpoint(self, f, next_epoch, best_running_loss=np.inf, streak=0):
        """Full training state after `next_epoch` epochs: parameters, BatchNorm statistics, both Adam moments, the
        device step / RNG counters, the host sampler's RNG, the early-stop bookkeeping and the loss history.  Rank-local:
        in a distributed run with the sharded optimiser every rank calls `engine.flush(collective=True)` first (the training
        loop does), else this raises instead of starting collectives on one rank."""
        eng, m = self.engine, self.model
        eng.flush()
        torch.save({'format': 'jamie_amd.ckpt.v2', 'epoch': int(next_epoch), 'input_dim': list(m.input_dim),
                    'output_dim': m.output_dim, 'dropout': m.dropout, 'batch_size': eng.B,
                    'compute_dtype': eng.compute_dtype,
                    'state_dict': {k: v.cpu() for k, v in m.state_dict().items()},
                    'flat': m.flat.cpu(), 'exp_avg': eng.exp_avg.cpu(), 'exp_avg_sq': eng.exp_avg_sq.cpu(),
                    'engine_state': eng.state.cpu(), 'num_batches_tracked': int(m.num_batches_tracked),
                    'best_running_loss': float(best_running_loss), 'streak': int(streak),
                    'loss_history': {k: list(v) for k, v in self.loss_history.items()},
                    'np_random': np.random.get_state(), 'py_random': random.getstate()}, f)

    def _load_checkpoint(self, f, eng):
        ck = torch.load(f, weights_only=False)
        m = self.model
        if ck.get('format') == 'jamie_amd.ckpt.v1':
            raise ValueError(f'{f}: checkpoint format v1 (round 2) is not supported after the parameter-layout change of '
                             f'format v2 (the flat buffers are ordered differently); re-train, or load its state_dict with load_model')
        if ck.get('format') != 'jamie_amd.ckpt.v2':
            raise ValueError(f'{f}: not a jamie_amd training checkpoint')
        if list(ck['input_dim']) != list(m.input_dim) or ck['output_dim'] != m.output_dim or ck['batch_size'] != eng.B \
                or ck['compute_dtype'] != eng.compute_dtype:
            raise ValueError(f'{f}: checkpoint of a different configuration (features {ck["input_dim"]}, latent '
                             f'{ck["output_dim"]}, batch {ck["batch_size"]}, {ck["compute_dtype"]})')
        m.load_state_dict(ck['state_dict'])
        m.flat.copy_(ck['flat'])                       # incl. alignment padding: bit-identical continuation
        m.num_batches_tracked = ck['num_batches_tracked']
        eng.exp_avg.copy_(ck['exp_avg'])
        eng.exp_avg_sq.copy_(ck['exp_avg_sq'])
        eng.state[1:].copy_(ck['engine_state'][1:])    # step counters; the Philox seed stays this rank's own
        if eng.bf16:
            eng.refresh_weights_bf16()
        if self.record_loss:
            self.loss_history = {k: list(v) for k, v in ck['loss_history'].items()}
        np.random.set_state(ck['np_random'])
        random.setstate(ck['py_random'])
        return ck['epoch'], ck['best_running_loss'], ck['streak']

    def load_model(self, f):
        """reference jamie.py:970-972."""
        ck = torch.load(f, weights_only=False)
        m = self.model_class(ck['input_dim'], ck['output_dim'], preprocessing=ck['preprocessing'],
                             preprocessing_inverse=ck['preprocessing_inverse'], dropout=ck['dropout'])
        m.load_state_dict(ck['state_dict'])
        self.model = m.to(self.device)
        self.dataset_num = self.model.num_modalities

    # ---- metrics kept for sanity checks (reference jamie.py:892-915) ----
    def test_closer(self, integrated_data, distance_metric=None):
        """FOSCTTM: fraction of samples closer than the true match."""
        assert len(integrated_data) == 2, 'Two datasets are supported for FOSCTTM'
        from sklearn.metrics.pairwise import pairwise_distances
        d = pairwise_distances(np.concatenate(integrated_data, axis=0), metric='euclidean')
        size = integrated_data[0].shape[0]
        closer = 0
        for i in range(size):
            ld = d[i][size:]
            closer += np.sum(ld < ld[i])
            ld = d[size + i][:size]
            closer += np.sum(ld < ld[i])
        foscttm = closer / (2 * size ** 2)
        print(f'foscttm: {foscttm}')
        return foscttm
