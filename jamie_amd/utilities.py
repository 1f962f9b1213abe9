"""Host-side helpers kept in Python, mirroring the reference's `jamie/utilities.py` for the hot path:
`preclass` (utilities.py:654-678), `identity` (:48-50) and `time_logger` (:61-132)."""
import tracemalloc
from time import perf_counter
import warnings

import numpy as np

from .model import identity  # noqa: F401  (re-export)


class preclass:
    """Standardise with the statistics of the fitting sample (reference utilities.py:654-678):
    `axis=0` -> per feature, `axis=None` -> one global mean/std (after PCA); NaN -> 0; and the inverse.
    The statistics are computed once (the reference recomputes them from the stored sample each call)."""

    def __init__(self, sample, pca=None, axis=None):
        self.pca = pca
        self.axis = axis
        self.mean = np.asarray(sample.mean(axis))
        self.std = np.asarray(sample.std(axis))

    @classmethod
    def from_stats(cls, mean, std, axis=0, pca=None):
        """The same object from statistics computed elsewhere (the device path: `_native.standardise_columns`, or
        `pca.global_standardise` behind a `pca.DevicePCA`)."""
        self = cls.__new__(cls)
        self.pca, self.axis = pca, axis
        self.mean, self.std = np.asarray(mean), np.asarray(std)
        return self

    def transform(self, X):
        out = X
        if self.pca is not None:
            out = self.pca.transform(out)
        out = out - self.mean
        with warnings.catch_warnings(), np.errstate(all='ignore'):
            warnings.simplefilter('ignore')
            out = out / self.std
        out[np.isnan(out)] = 0
        return out

    def inverse_transform(self, X):
        out = np.asarray(X)
        out = out * self.std
        out = out + self.mean
        if self.pca is not None:
            out = self.pca.inverse_transform(out)
        return out


class time_logger:
    """Phase timer with the reference's interface and labels (utilities.py:61-132).  `sync` is called
    before each reading so that phases cover the GPU work they launched (HIP launches are asynchronous).
    `memory_usage=True` (utilities.py:78-80, 98-111, 123-130): per phase the host memory traced by `tracemalloc`
    (stored, peak) as in the reference, and next to it the device memory of this process (HBM allocated now / peak
    since the previous reading) in `history_mem_device`."""

    def __init__(self, discard_first_sample=False, record=True, verbose=False, memory_usage=False, sync=None):
        self.discard_first_sample = discard_first_sample
        self.record = record
        self.verbose = verbose
        self.memory_usage = memory_usage
        self.sync = sync
        self.history = {}
        if memory_usage:
            self.history_mem, self.history_mem_device = {}, {}
            tracemalloc.start()
        self.start_time = perf_counter()

    @staticmethod
    def _device_memory():
        try:
            import torch
            if not torch.cuda.is_available():
                return (0, 0)
            now, peak = torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()
            torch.cuda.reset_peak_memory_stats()
            return (now, peak)
        except Exception:
            return (0, 0)

    def log(self, str=''):
        if not (self.verbose or self.record):
            return
        if self.sync is not None:
            self.sync()
        self.end_time = perf_counter()
        elapsed = self.end_time - self.start_time
        if self.record:
            self.history.setdefault(str, []).append(elapsed)
        if self.verbose:
            print(f'{str}: {elapsed}')
        if self.memory_usage:
            host, device = tracemalloc.get_traced_memory(), self._device_memory()
            if self.record:
                self.history_mem.setdefault(str, []).append(host)
                self.history_mem_device.setdefault(str, []).append(device)
            if self.verbose:
                print(f'{str} Memory: Stored {host[0]} - Peak {host[1]} (device: allocated {device[0]} - peak {device[1]})')
            tracemalloc.stop()
            tracemalloc.start()               # per-phase figures, like the reference
        self.start_time = perf_counter()

    def aggregate(self):
        running_total = 0
        for k, v in self.history.items():
            v = np.array(v[1:] if (self.discard_first_sample and len(v) > 1) else v)
            running_total += v.mean()
            print(f'{k}: {v.mean()}')
            if self.memory_usage and k in self.history_mem:
                stored = sum(val[0] for val in self.history_mem[k])
                peak = max(val[1] for val in self.history_mem[k])
                dpeak = max(val[1] for val in self.history_mem_device[k])
                print(f'{k} Memory: Stored {stored} - Peak {peak} (device peak {dpeak})')
        print(f'Total: {running_total}')


def geodesic_distances(X, kmax):
    """`unioncom.utils.geodesic_distances` (unioncom 0.4.0, absent from /root/reference and from this image), restated
    from its published algorithm -- PARITY UNPINNED: k-nearest-neighbour graph (k from 5, +2 until the graph is
    connected or k exceeds max(kmax, 1 % of the cells)), all-pairs shortest paths, unreachable pairs set to twice the
    largest finite distance.  Host side (scipy / sklearn), like every distance mode of the reference
    (jamie.py:839-890)."""
    import scipy.sparse.csgraph as csgraph
    from sklearn.neighbors import NearestNeighbors
    X = np.asarray(X)
    kmin = 5

    def graph(k):
        nbrs = NearestNeighbors(n_neighbors=min(k, len(X)), metric='euclidean').fit(X)
        return nbrs.kneighbors_graph(X, mode='distance')
    knn = graph(kmin)
    while csgraph.connected_components(knn, directed=False)[0] != 1:
        if kmin > np.max((kmax, 0.01 * len(X))):
            break
        kmin += 2
        knn = graph(kmin)
    dist = csgraph.shortest_path(knn, method='D', directed=False)
    finite = dist[np.isfinite(dist)]
    dist_max = finite.max() if finite.size else 0.0
    dist[~np.isfinite(dist)] = 2 * dist_max
    return dist


def distance_matrix(X, mode, kmax=40):
    """One modality's cell x cell distance matrix, reference jamie.py:839-890 (`compute_distances`)."""
    from scipy import stats
    from sklearn.metrics import pairwise_distances
    if mode == 'geodesic':                                                      # jamie.py:851-856
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return np.array(geodesic_distances(X, kmax))
    if mode == 'spearman':                                                      # jamie.py:857-869
        if X.shape[0] == 1:
            return np.array([0])
        d, _ = stats.spearmanr(X, axis=1)
        if np.isnan(d).any():
            raise Exception('Data is not well conditioned for spearman method '
                            '(scipy.stats.spearmanr returned ``np.nan``)')
        if len(np.shape(d)) == 0:
            d = np.array([[1, d], [d, 1]])
        return (1 - np.array(d)) / 2
    if mode == 'pearson':                                                       # jamie.py:870-879
        if X.shape[0] == 1:
            return np.array([0])
        d = np.corrcoef(X.toarray() if hasattr(X, 'toarray') else np.asarray(X))
        if len(np.shape(d)) == 0:
            d = np.array([[1, d], [d, 1]])
        return (1 - np.array(d)) / 2
    return pairwise_distances(X, metric=mode)                                   # jamie.py:880-882
