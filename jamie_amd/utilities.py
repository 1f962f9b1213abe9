"""Host-side helpers kept in Python, mirroring the reference's `jamie/utilities.py` for the hot path:
`preclass` (utilities.py:654-678), `identity` (:48-50) and `time_logger` (:61-132)."""
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
    before each reading so that phases cover the GPU work they launched (HIP launches are asynchronous)."""

    def __init__(self, discard_first_sample=False, record=True, verbose=False, memory_usage=False, sync=None):
        self.discard_first_sample = discard_first_sample
        self.record = record
        self.verbose = verbose
        self.memory_usage = memory_usage
        self.sync = sync
        self.history = {}
        self.start_time = perf_counter()

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
        self.start_time = perf_counter()

    def aggregate(self):
        running_total = 0
        for k, v in self.history.items():
            v = np.array(v[1:] if (self.discard_first_sample and len(v) > 1) else v)
            running_total += v.mean()
            print(f'{k}: {v.mean()}')
        print(f'Total: {running_total}')
