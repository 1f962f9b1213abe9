"""jamie_amd — MI355X-native (gfx950) implementation of JAMIE's coupled-VAE training / inference hot path.

Drop-in surface (reference Oafish1/JAMIE v4.4.5, jamie/jamie.py): `JAMIE(...).fit_transform()`,
`.transform()`, `.transform_one()`, `.modal_predict()`, `.save_model()`, `.load_model()`, plus the
`fit()` / `impute()` spellings.  The compute path is hand-written HIP reached through the C ABI of
`libjamie_hip.so` (include/jamie_hip.h); there is no CPU fallback.
"""
__version__ = '0.1.0'

from .build import build_experiments, build_library, library_path  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require the GPU library (build() imports it first)
    if name in ('JAMIE',):
        from .jamie import JAMIE
        return JAMIE
    if name in ('edModelVar', 'ParamLayout'):
        from . import model
        return getattr(model, name)
    if name in ('edModelVarTorch',):
        from .compat import edModelVarTorch
        return edModelVarTorch
    if name in ('TrainEngine',):
        from .engine import TrainEngine
        return TrainEngine
    raise AttributeError(name)
