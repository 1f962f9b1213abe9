"""ctypes binding of libjamie_hip.so (C ABI declared in include/jamie_hip.h).

There is NO CPU fallback: importing this module without the built library, or calling an op without a
GPU, raises.  PyTorch is used only for device memory and streams; every op below receives raw device
pointers and launches on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# JAMIE_HIP_LIB: load a diagnostic build (tools/stamp_gemm_bf16.sh) instead of the in-tree product library
LIB_PATH = os.environ.get('JAMIE_HIP_LIB') or os.path.join(_HERE, 'libjamie_hip.so')

MAX_GROUP = 4
MAX_GEMM_GROUP = 8          # JAMIE_MAX_GEMM_GROUP: problems per grouped GEMM launch
MAX_GEMM_GROUP_F32 = 12     # JAMIE_MAX_GEMM_GROUP_F32: ... of the fp32 entry points
NT, NN, TN = 0, 1, 2
EPI_STORE, EPI_MSE, EPI_BN_EVAL = 0, 1, 2

c_f32p = C.c_void_p


class GemmProblem(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('bias', C.c_void_p),
                ('aux0', C.c_void_p), ('aux1', C.c_void_p), ('aux2', C.c_void_p), ('aux3', C.c_void_p),
                ('partial', C.c_void_p), ('a_rows', C.c_void_p),
                ('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
                ('lda', C.c_int), ('ldb', C.c_int), ('ldc', C.c_int), ('aux_ld', C.c_int),
                ('splitk', C.c_int), ('slab_stride', C.c_longlong),
                ('epi', C.c_int), ('accumulate', C.c_int),
                ('scale', C.c_float), ('slope', C.c_float), ('eps', C.c_float), ('pscale', C.c_float), ('b_tr', C.c_int), ('a_tr', C.c_int), ('store_nt', C.c_int), ('c_bf16', C.c_int), ('c_panel', C.c_int)]


class CastProblem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('dstT', C.c_void_p),
                ('R', C.c_int), ('C', C.c_int), ('ld', C.c_int), ('ldd', C.c_int), ('ldt', C.c_int),
                ('nslab', C.c_int), ('slab_stride', C.c_longlong), ('src_bf16', C.c_void_p),
                ('rows', C.c_void_p), ('dst32', C.c_void_p), ('ld32', C.c_int)]


class MseProblem(C.Structure):
    _fields_ = [('y', C.c_void_p), ('x', C.c_void_p), ('d', C.c_void_p), ('d_bf16', C.c_void_p), ('dT_bf16', C.c_void_p),
                ('partial', C.c_void_p), ('R', C.c_int), ('C', C.c_int), ('nslab', C.c_int),
                ('slab_stride', C.c_longlong), ('scale', C.c_float), ('pscale', C.c_float), ('colpart', C.c_void_p)]


class ColsumProblem(C.Structure):
    _fields_ = [('X', C.c_void_p), ('out', C.c_void_p), ('M', C.c_int), ('N', C.c_int), ('ld', C.c_int),
                ('nslab', C.c_int), ('slab_stride', C.c_longlong), ('accumulate', C.c_int)]


class BnFwdProblem(C.Structure):
    _fields_ = [('h', C.c_void_p), ('nslab', C.c_int), ('slab_stride', C.c_longlong),
                ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('save_mean', C.c_void_p), ('save_invstd', C.c_void_p),
                ('out', C.c_void_p), ('mask', C.c_void_p),
                ('B', C.c_int), ('N', C.c_int), ('rng_stream', C.c_int),
                ('out_bf16', C.c_void_p), ('outT_bf16', C.c_void_p), ('panel', C.c_int)]


class BnBwdProblem(C.Structure):
    _fields_ = [('da', C.c_void_p), ('nslab', C.c_int), ('slab_stride', C.c_longlong),
                ('h', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('save_mean', C.c_void_p), ('save_invstd', C.c_void_p),
                ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('dbias_lin', C.c_void_p),
                ('mask', C.c_void_p),
                ('B', C.c_int), ('N', C.c_int), ('rng_stream', C.c_int), ('accumulate', C.c_int),
                ('dh_bf16', C.c_void_p), ('dhT_bf16', C.c_void_p), ('skip_f32', C.c_int), ('panel', C.c_int)]


class Latent(C.Structure):
    _fields_ = [('B', C.c_int), ('L', C.c_int),
                ('ml', C.c_void_p * 2), ('ml_nslab', C.c_int), ('ml_slab_stride', C.c_longlong),
                ('head_bias', C.c_void_p * 2), ('eps_in', C.c_void_p * 2),
                ('sigma', C.c_void_p), ('corr', C.c_void_p), ('Fblk', C.c_void_p), ('hyper', C.c_void_p),
                ('mu', C.c_void_p * 2), ('lv', C.c_void_p * 2), ('z', C.c_void_p * 2),
                ('eps', C.c_void_p * 2), ('comb', C.c_void_p * 2), ('cz', C.c_void_p * 2),
                ('rsum', C.c_void_p), ('qsum', C.c_void_p), ('fc1', C.c_void_p), ('partials', C.c_void_p),
                ('dcomb', C.c_void_p * 2), ('dcomb_nslab', C.c_int), ('dcomb_slab_stride', C.c_longlong),
                ('H', C.c_void_p * 2), ('ch', C.c_void_p * 2), ('fte', C.c_void_p),
                ('dml', C.c_void_p * 2), ('dsigma', C.c_void_p),
                ('rec_partials', C.c_void_p), ('n_rec_partials', C.c_int), ('losses', C.c_void_p),
                ('cosine', C.c_int), ('rng_stream', C.c_int),
                ('dz_ext', C.c_void_p * 2), ('dmu_ext', C.c_void_p * 2), ('dlv_ext', C.c_void_p),
                ('comb_bf16', C.c_void_p * 2), ('combT_bf16', C.c_void_p * 2),
                ('dml_bf16', C.c_void_p * 2), ('dmlT_bf16', C.c_void_p * 2)]


class LatentM(C.Structure):
    _fields_ = [('B', C.c_int), ('L', C.c_int), ('M', C.c_int),
                ('ml', C.c_void_p * 4), ('ml_nslab', C.c_int), ('ml_slab_stride', C.c_longlong),
                ('head_bias', C.c_void_p * 4), ('eps_in', C.c_void_p * 4),
                ('sigma', C.c_void_p), ('hyper', C.c_void_p),
                ('mu', C.c_void_p * 4), ('lv', C.c_void_p * 4), ('z', C.c_void_p * 4), ('eps', C.c_void_p * 4),
                ('comb', C.c_void_p), ('partials', C.c_void_p),
                ('dcomb', C.c_void_p * 4), ('dcomb_nslab', C.c_int), ('dcomb_slab_stride', C.c_longlong),
                ('dml', C.c_void_p * 4), ('dsigma', C.c_void_p),
                ('rec_partials', C.c_void_p), ('n_rec_partials', C.c_int), ('losses', C.c_void_p),
                ('rng_stream', C.c_int),
                ('g1', C.c_void_p * 4), ('dec0_W', C.c_void_p * 4), ('dec0_b', C.c_void_p * 4), ('d', C.c_int * 4),
                ('comb_alias', C.c_void_p * 4), ('comb_bf16', C.c_void_p * 4), ('combT_bf16', C.c_void_p * 4),
                ('dml_bf16', C.c_void_p * 4), ('dmlT_bf16', C.c_void_p * 4),
                ('dbias_head', C.c_void_p * 4), ('colpart', C.c_void_p), ('accumulate', C.c_int), ('ticket', C.c_void_p),
                ('defer_final', C.c_int), ('head_W', C.c_void_p * 4), ('da2', C.c_void_p * 4), ('dec0_WT_bf16', C.c_void_p * 4),
                ('g1_panel', C.c_int), ('da2_panel', C.c_int), ('heads_a_bf16', C.c_void_p * 4), ('heads_W_bf16', C.c_void_p * 4)]


class SampleArgs(C.Structure):
    _fields_ = [('idx', C.c_void_p), ('B', C.c_int), ('N', C.c_longlong), ('offset', C.c_longlong), ('replace', C.c_int),
                ('rng_stream', C.c_int), ('step_add', C.c_int)]


class PdState(C.Structure):
    _fields_ = [('F', C.c_void_p), ('G1', C.c_void_p), ('G2', C.c_void_p), ('m1', C.c_void_p), ('m2', C.c_void_p),
                ('Mu', C.c_void_p), ('Lambda', C.c_void_p), ('S', C.c_void_p), ('rowsum', C.c_void_p),
                ('colsum', C.c_void_p), ('alpha', C.c_void_p), ('rowpart', C.c_void_p), ('colpart', C.c_void_p),
                ('m', C.c_int), ('n', C.c_int), ('rho', C.c_float), ('epsilon', C.c_float)]


EXPORTS = {
    'jamie_last_error': (C.c_char_p, []),
    'jamie_version': (C.c_int, []),
    'jamie_panel_width': (C.c_int, []),
    'jamie_max_partials': (C.c_int, []),
    'jamie_max_norm_partials': (C.c_int, []),
    'jamie_gemm_f32': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_int, C.c_void_p]),
    'jamie_gemm_f32_cfg': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'jamie_gemm_tile': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'jamie_gemm_bf16': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_int, C.c_void_p]),
    'jamie_gemm_bf16_ranges': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.c_void_p, C.POINTER(LatentM), C.c_void_p]),
    'jamie_gemm_bf16_tile': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'jamie_comm_version': (C.c_int, [C.POINTER(C.c_int)]),
    'jamie_comm_unique_id': (C.c_int, [C.c_void_p]),
    'jamie_comm_create': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'jamie_comm_destroy': (C.c_int, [C.c_void_p]),
    'jamie_allreduce': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    'jamie_reduce_scatter': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    'jamie_all_gather': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    'jamie_comm_wait': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'jamie_cast_transpose': (C.c_int, [C.POINTER(CastProblem), C.c_int, C.c_void_p]),
    'jamie_mse_cast': (C.c_int, [C.POINTER(MseProblem), C.c_int, C.c_void_p]),
    'jamie_bn_act_fwd': (C.c_int, [C.POINTER(BnFwdProblem), C.c_int, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_void_p, C.c_void_p]),
    'jamie_bn_act_bwd': (C.c_int, [C.POINTER(BnBwdProblem), C.c_int, C.c_float, C.c_float, C.c_void_p,
                                   C.c_void_p]),
    'jamie_bn_act_bwd_cs': (C.c_int, [C.POINTER(BnBwdProblem), C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    'jamie_bn_act_fwd_pf': (C.c_int, [C.POINTER(BnFwdProblem), C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'jamie_bn_act_bwd_pf': (C.c_int, [C.POINTER(BnBwdProblem), C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'jamie_latent_fwd': (C.c_int, [C.POINTER(Latent), C.c_void_p, C.c_void_p]),
    'jamie_latent_bwd': (C.c_int, [C.POINTER(Latent), C.c_void_p]),
    'jamie_latent_m_fwd': (C.c_int, [C.POINTER(LatentM), C.c_void_p, C.c_void_p]),
    'jamie_latent_m_bwd': (C.c_int, [C.POINTER(LatentM), C.c_void_p]),
    'jamie_latent_m_bwd_ex': (C.c_int, [C.POINTER(LatentM), C.POINTER(SampleArgs), C.c_void_p, C.c_void_p]),
    'jamie_latent_m_colpart_size': (C.c_longlong, [C.c_int, C.c_int]),
    'jamie_optim_blocks': (C.c_int, [C.c_longlong]),
    'jamie_grad_sqnorm': (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'jamie_grad_sqnorm_bf16': (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'jamie_clip_adam_g16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'jamie_clip_adam_ride': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SampleArgs),
                                       C.POINTER(CastProblem), C.c_int, C.c_void_p]),
    'jamie_grad_sqnorm_ranges': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    'jamie_grad_sqnorm_ranges_g16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    'jamie_grad_sqnorm_ranges_fin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                               C.c_void_p, C.POINTER(LatentM), C.c_void_p, C.c_int, C.c_void_p]),
    'jamie_sqnorm_range_blocks': (C.c_int, [C.c_void_p, C.c_int]),
    'jamie_clip_adam': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'jamie_dense_block': (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong,
                                    C.c_longlong, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    'jamie_axpby': (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_longlong, C.c_void_p]),
    'jamie_gather_rows': (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    'jamie_sample_indices_group': (C.c_int, [C.POINTER(SampleArgs), C.c_int, C.c_void_p, C.c_void_p]),
    'jamie_sample_indices': (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p,
                                       C.c_int, C.c_void_p]),
    'jamie_hybrid_assemble': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'jamie_corr_from_indices': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'jamie_csr_block': (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    'jamie_colsum_group': (C.c_int, [C.POINTER(ColsumProblem), C.c_int, C.c_void_p]),
    'jamie_colsum': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p,
                               C.c_int, C.c_void_p]),
    'jamie_col_stats': (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_longlong, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    'jamie_standardise': (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    'jamie_pd_workspace': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'jamie_pd_step': (C.c_int, [C.POINTER(PdState), C.c_int, C.c_void_p]),
    'jamie_pd_alpha': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                 C.c_void_p]),
}

# entry points of the EXPERIMENTS build only (libjamie_hip_exp.so: jamie_amd/experiments.py binds them when the loaded library has them)
EXPERIMENT_EXPORTS = {
    'jamie_gemm_bf16_ring_plan': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'jamie_gemm_bf16_ring': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(LatentM), C.c_void_p, C.c_void_p]),
    'jamie_gemm_bf16_skinny': (C.c_int, [C.POINTER(GemmProblem), C.c_int, C.c_void_p]),
    'jamie_gemm_bf16_bn': (C.c_int, [C.POINTER(GemmProblem), C.POINTER(BnFwdProblem), C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


class JamieHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and declare every exported symbol's signature."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('JAMIE_LIB') or LIB_PATH          # JAMIE_LIB: an A/B build of the same sources (tools/ab.sh)
    if not os.path.exists(path):
        raise JamieHipError(
            f'{path} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  jamie_amd has no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENT_EXPORTS.items():
        fn = getattr(lib, name, None)    # (present in an experiments build only)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise JamieHipError(f'libjamie_hip error {rc}: {load().jamie_last_error().decode()}')


# ---------------------------------------------------------------------------------------------------
# launch plans: the training step is a fixed sequence of C-ABI calls on static buffers, so it is recorded once
# and replayed with one foreign call per launch (building the ctypes descriptors costs ~10x the call itself).
# ---------------------------------------------------------------------------------------------------
_rec = None


def begin_record():
    global _rec
    _rec = []


def end_record():
    global _rec
    plan, _rec = _rec, None
    return plan


def record_callable(fn):
    """Insert a Python callable (event record, collective, ...) into the plan being recorded; returns True if
    a recording is active (the caller should then NOT run it itself unless it wants it executed now too)."""
    if _rec is not None:
        _rec.append((fn, None))
        return True
    return False


def replay(plan):
    global _stream_override
    st = _stream()
    lib_err = None
    try:
        for fn, args in plan:
            if fn is _SET_STREAM:                # later entries launch on another HIP stream (None = torch's current)
                _stream_override = args
                st = _stream()
            elif args is None:
                fn()
            else:
                rc = fn(*args[:-1], st)          # the stream is the last argument of every launch entry point
                if rc:
                    lib_err = rc
    finally:
        _stream_override = None
    if lib_err:
        _check(lib_err)


def _call(name, *args):
    fn = getattr(load(), name)
    if _rec is not None:
        _rec.append((fn, args))
    _check(fn(*args))


# launches go to torch's current HIP stream unless `set_stream` names another one (the optimiser stream of the
# pipelined step, engine.TrainEngine.enable_pipeline); stream switches are plan entries too
_SET_STREAM = object()
_stream_override = None


def set_stream(stream):
    """Launch the following ops on `stream` (a torch.cuda.Stream; None = back to torch's current stream)."""
    global _stream_override
    _stream_override = stream
    if _rec is not None:
        _rec.append((_SET_STREAM, stream))


def current_stream():
    return _stream_override if _stream_override is not None else torch.cuda.current_stream()


def _stream():
    return C.c_void_p(current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must already live on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise JamieHipError('jamie_amd ops need GPU tensors (no CPU fallback)')
    return t.data_ptr()


def require_gpu():
    if not torch.cuda.is_available():
        raise JamieHipError('no MI355X visible: jamie_amd runs on the GPU only (no CPU fallback)')
    load()


# ---------------------------------------------------------------------------------------------------
# thin op wrappers
# ---------------------------------------------------------------------------------------------------
def gemm_problem(A, B, Cout, M, N, K, lda, ldb, ldc, bias=None, splitk=1, slab_stride=0, epi=EPI_STORE,
                 accumulate=False, aux=(None, None, None, None), aux_ld=0, partial=None, a_rows=None,
                 scale=1.0, slope=0.01, eps=1e-5, pscale=1.0, b_tr=False, a_tr=False, store_nt=False, c_bf16=False, c_panel=False):
    p = GemmProblem()
    p.b_tr, p.a_tr, p.store_nt, p.c_bf16, p.c_panel = int(b_tr), int(a_tr), int(store_nt), int(c_bf16), int(c_panel)
    p.A, p.B, p.C, p.bias = ptr(A), ptr(B), ptr(Cout), ptr(bias)
    p.aux0, p.aux1, p.aux2, p.aux3 = (ptr(a) for a in aux)
    p.partial, p.a_rows = ptr(partial), ptr(a_rows)
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc, p.aux_ld = M, N, K, lda, ldb, ldc, aux_ld
    p.splitk, p.slab_stride, p.epi, p.accumulate = splitk, slab_stride, epi, int(accumulate)
    p.scale, p.slope, p.eps, p.pscale = scale, slope, eps, pscale
    p._keep = (A, B, Cout, bias, aux, partial, a_rows)      # keep the tensors alive with the descriptor
    return p


def gemm(problems, layout, cfg=-1):
    arr = (GemmProblem * len(problems))(*problems)
    _call('jamie_gemm_f32_cfg', arr, len(problems), layout, cfg, _stream())


def gemm_tile(layout, max_m, max_n, max_k, cfg=-1):
    bm, bn = C.c_int(), C.c_int()
    _check(load().jamie_gemm_tile(layout, max_m, max_n, max_k, cfg, C.byref(bm), C.byref(bn)))
    return bm.value, bn.value


def gemm_bf16(problems, cfg=-1, ranges=None):
    """`ranges` = (g, g16 or None, SqRanges, partials, state, fin or None): the range-norm work (grad_sqnorm_ranges) rides as
    extra workgroups of this launch (tile configuration 29: the last dW launch of a backward pass)."""
    arr = (GemmProblem * len(problems))(*problems)
    if ranges is not None:
        g, g16, rg, partials, state, fin = ranges
        _call('jamie_gemm_bf16_ranges', arr, len(problems), cfg, ptr(g), ptr(g16), rg.off, rg.len, rg.count, ptr(partials),
              partials.numel(), ptr(state), C.pointer(fin) if fin is not None else None, _stream())
        return
    _call('jamie_gemm_bf16', arr, len(problems), cfg, _stream())


# ---- RCCL collectives behind the C ABI (csrc/comm.hip): one foreign call per collective, recordable in a launch plan ----
def _comm_dtype(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise JamieHipError(f'collectives take fp32 or bf16 buffers, not {t.dtype}')


def comm_unique_id():
    """128-byte RCCL unique id (bytes): made by rank 0, handed to every rank's comm_create."""
    buf = (C.c_char * 128)()
    _check(load().jamie_comm_unique_id(buf))
    return bytes(buf)


def comm_create(uid, rank, world):
    h = C.c_void_p()
    _check(load().jamie_comm_create(C.c_char_p(uid), int(rank), int(world), C.byref(h)))
    return h


def comm_destroy(h):
    if h:
        load().jamie_comm_destroy(h)


def comm_version():
    v = C.c_int()
    _check(load().jamie_comm_version(C.byref(v)))
    return v.value


def comm_allreduce(h, buf, slot):
    _call('jamie_allreduce', h, ptr(buf), buf.numel(), _comm_dtype(buf), int(slot), _stream())


def comm_reduce_scatter(h, send, recv, slot):
    _call('jamie_reduce_scatter', h, ptr(send), ptr(recv), recv.numel(), _comm_dtype(recv), int(slot), _stream())


def comm_all_gather(h, send, recv, slot):
    _call('jamie_all_gather', h, ptr(send), ptr(recv), send.numel(), _comm_dtype(send), int(slot), _stream())


def comm_wait(h, slot):
    _call('jamie_comm_wait', h, int(slot), _stream())


def gemm_bf16_tile(max_m, max_n, cfg=-1):
    bm, bn = C.c_int(), C.c_int()
    _check(load().jamie_gemm_bf16_tile(max_m, max_n, cfg, C.byref(bm), C.byref(bn)))
    return bm.value, bn.value


def cast_problem(src, dst=None, dstT=None, nslab=1, slab_stride=0, rows=None, dst32=None):
    """fp32 [R, C] (contiguous 2-D view; slabs `slab_stride` elements apart) -> bf16 dst [R, C] / dstT [C, R].
    `rows` (int32 [R'] device tensor): output row r reads source row rows[r]; `dst32`: fp32 copy of the output rows."""
    p = CastProblem()
    R, Cc = src.shape[-2], src.shape[-1]
    if rows is not None:
        p.rows, R = ptr(rows), rows.numel()
    if dst32 is not None:
        p.dst32, p.ld32 = ptr(dst32), Cc
    if src.dtype == torch.bfloat16:      # bf16 source: transposed (or plain) copy of an existing bf16 matrix
        p.src_bf16, p.dst, p.dstT = ptr(src), ptr(dst), ptr(dstT)
    else:
        p.src, p.dst, p.dstT = ptr(src), ptr(dst), ptr(dstT)
    p.R, p.C, p.ld, p.ldd, p.ldt, p.nslab, p.slab_stride = R, Cc, Cc, Cc, R, nslab, slab_stride
    p._keep = (src, dst, dstT, rows, dst32)
    return p


def mse_problem(y, x, d, d_bf16=None, dT_bf16=None, partial=None, scale=1.0, pscale=1.0, colpart=None):
    """y [nslab, R, C] fp32 slabs (contiguous), x [R, C] fp32; outputs: d [R, C] fp32 (or None), optional bf16 [R, C] / [C, R]
    copies of d, optional `colpart` [ceil(R / 64), C] (column sums of d per 64-row tile)."""
    p = MseProblem()
    nslab = y.shape[0] if y.dim() == 3 else 1
    R, Cc = x.shape
    p.y, p.x, p.d, p.d_bf16, p.dT_bf16, p.partial = ptr(y), ptr(x), ptr(d), ptr(d_bf16), ptr(dT_bf16), ptr(partial)
    p.R, p.C, p.nslab, p.slab_stride, p.scale, p.pscale = R, Cc, nslab, R * Cc, scale, pscale
    p.colpart = ptr(colpart)
    p._keep = (y, x, d, d_bf16, dT_bf16, partial, colpart)
    return p


def mse_cast(problems):
    arr = (MseProblem * len(problems))(*problems)
    _call('jamie_mse_cast', arr, len(problems), _stream())


def cast_transpose(problems):
    for i in range(0, len(problems), 16):
        chunk = problems[i:i + 16]
        arr = (CastProblem * len(chunk))(*chunk)
        _call('jamie_cast_transpose', arr, len(chunk), _stream())


class FlatCast:
    """fp32 -> bf16 copy of a contiguous range (the data-parallel message buffer): `jamie_cast_transpose` on the range
    viewed as [R, 2048] plus a tail row; the descriptors are built once, `run(stream)` is one foreign call and is never
    recorded into a launch plan (its caller, the gradient exchange, is itself a plan entry)."""

    def __init__(self, src, dst, width=2048):
        n = src.numel()
        assert dst.numel() == n and src.dtype == torch.float32 and dst.dtype == torch.bfloat16
        probs = []
        R = n // width
        if R:
            probs.append(cast_problem(src[:R * width].view(R, width), dst[:R * width].view(R, width)))
        if n - R * width:
            probs.append(cast_problem(src[R * width:].view(1, -1), dst[R * width:].view(1, -1)))
        self.arr = (CastProblem * len(probs))(*probs)
        self.n = len(probs)
        self._keep = probs
        self.fn = load().jamie_cast_transpose

    def run(self, stream):
        _check(self.fn(self.arr, self.n, C.c_void_p(stream.cuda_stream)))


def _pf_arrays(tensors):
    """Host arrays (device pointers, byte counts rounded down to 16) of up to 8 contiguous tensors for a prefetch rider."""
    pp = (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])
    pb = (C.c_longlong * len(tensors))(*[t.numel() * t.element_size() // 16 * 16 for t in tensors])
    return pp, pb


def bn_act_fwd(problems, p_drop, rng, momentum=0.1, eps=1e-5, slope=0.01, prefetch=None):
    """`prefetch` (list of <= 8 contiguous tensors): extra workgroups read them into the caches for the next launches
    (jamie_bn_act_fwd_pf)."""
    arr = (BnFwdProblem * len(problems))(*problems)
    if prefetch:
        pp, pb = _pf_arrays(prefetch)
        arr._keep = (prefetch, pp, pb)
        _call('jamie_bn_act_fwd_pf', arr, len(problems), p_drop, momentum, eps, slope, ptr(rng), pp, pb, len(prefetch), _stream())
        return
    _call('jamie_bn_act_fwd', arr, len(problems), p_drop, momentum, eps, slope, ptr(rng), _stream())


def bn_act_bwd(problems, p_drop, rng, slope=0.01, colsums=None, prefetch=None):
    """`colsums` (from colsum_problems): column sums computed by extra workgroups of the same launch; `prefetch`: as bn_act_fwd."""
    arr = (BnBwdProblem * len(problems))(*problems)
    if prefetch:
        pp, pb = _pf_arrays(prefetch)
        arr._keep = (prefetch, pp, pb)
        _call('jamie_bn_act_bwd_pf', arr, len(problems), p_drop, slope, ptr(rng), colsums[0] if colsums is not None else None,
              colsums[1] if colsums is not None else 0, pp, pb, len(prefetch), _stream())
        return
    if colsums is not None:
        _call('jamie_bn_act_bwd_cs', arr, len(problems), p_drop, slope, ptr(rng), colsums[0], colsums[1], _stream())
        return
    _call('jamie_bn_act_bwd', arr, len(problems), p_drop, slope, ptr(rng), _stream())


def latent_fwd(desc, rng):
    _call('jamie_latent_m_fwd' if isinstance(desc, LatentM) else 'jamie_latent_fwd', C.pointer(desc), ptr(rng), _stream())


def sample_args(idx, N, offset, replace, rng_stream, step_add=0):
    """What a riding sampler draws (jamie_sample_args): idx.numel() indices of [0, N) + offset into idx."""
    a = SampleArgs()
    a.idx, a.B, a.N, a.offset, a.replace, a.rng_stream, a.step_add = ptr(idx), idx.numel(), int(N), int(offset), int(replace), \
        int(rng_stream), int(step_add)
    a._keep = idx
    return a


def latent_bwd(desc, sample=None, state=None):
    """`sample` (sample_args, fused M-modality path only): the launch's extra workgroup draws the NEXT step's batch."""
    if sample is not None:
        _call('jamie_latent_m_bwd_ex', C.pointer(desc), C.pointer(sample), ptr(state), _stream())
        return
    _call('jamie_latent_m_bwd' if isinstance(desc, LatentM) else 'jamie_latent_bwd', C.pointer(desc), _stream())


def optim_blocks(n):
    return load().jamie_optim_blocks(n)


def grad_sqnorm(g, partials, state):
    name = 'jamie_grad_sqnorm_bf16' if g.dtype == torch.bfloat16 else 'jamie_grad_sqnorm'
    _call(name, ptr(g), g.numel(), ptr(partials), partials.numel(), ptr(state), _stream())


def clip_adam(p, g, m, v, partials, hyper, state, p_bf16=None, sample=None, casts=None):
    """`g`: the fp32 gradient, or (a bf16 tensor) the reduced gradient left in its bf16 message buffer.
    Extra workgroups of the same launch, one kind at most: `sample` (sample_args) draws the next step's batch into idx;
    `casts` (list of cast_problem, <= 16) gathers / casts the next step's batch rows."""
    if sample is not None or casts:
        arr = (CastProblem * len(casts))(*casts) if casts else None
        _call('jamie_clip_adam_ride', ptr(p), ptr(g), int(g.dtype == torch.bfloat16), ptr(m), ptr(v), p.numel(), ptr(partials),
              partials.numel(), ptr(hyper), ptr(state), ptr(p_bf16), C.pointer(sample) if sample is not None else None,
              arr, len(casts) if casts else 0, _stream())
        return
    name = 'jamie_clip_adam_g16' if g.dtype == torch.bfloat16 else 'jamie_clip_adam'
    _call(name, ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(partials), partials.numel(),
          ptr(hyper), ptr(state), ptr(p_bf16), _stream())


def gather_rows(src, idx, dst):
    _call('jamie_gather_rows', ptr(src), src.shape[0], src.shape[1], ptr(idx), idx.numel(), ptr(dst), _stream())


def sample_indices_group(items, rng):
    """items = [sample_args(...)] (<= 4): independent draws in ONE launch, one workgroup each."""
    arr = (SampleArgs * len(items))(*items)
    arr._keep = items
    _call('jamie_sample_indices_group', arr, len(items), ptr(rng), _stream())


def sample_indices(idx, N, offset, replace, rng, rng_stream):
    _call('jamie_sample_indices', ptr(idx), idx.numel(), N, offset, int(replace), ptr(rng), rng_stream, _stream())


def corr_from_indices(idx0, idx1, corr):
    _call('jamie_corr_from_indices', ptr(idx0), ptr(idx1), idx0.numel(), ptr(corr), _stream())


def colsum_group(items, accumulate=False):
    """items: list of (X [M, N] contiguous fp32, out [N]) - one launch for all (bias gradients of both modalities)."""
    probs = []
    for X, out in items:
        p = ColsumProblem()
        p.X, p.out, p.M, p.N, p.ld, p.nslab, p.slab_stride, p.accumulate = ptr(X), ptr(out), X.shape[0], X.shape[1], X.shape[1], 1, 0, int(accumulate)
        probs.append(p)
    arr = (ColsumProblem * len(probs))(*probs)
    _call('jamie_colsum_group', arr, len(probs), _stream())


def csr_block(indptr, indices, vals, idx0, idx1, out, row_off=0, col_off=0, normalise=True):
    """out[a, b] = P[idx0[a] + row_off, idx1[b] + col_off] for a device CSR matrix (sorted column indices)."""
    _call('jamie_csr_block', ptr(indptr), ptr(indices), ptr(vals), ptr(idx0), ptr(idx1), idx0.numel(), idx1.numel(),
          int(row_off), int(col_off), int(normalise), ptr(out), _stream())


def dense_block(Mat, idx0, idx1, out, row_off=0, col_off=0, normalise=True, w_blk=1.0, add=None, w_add=0.0):
    """out = w_blk * rownorm(Mat[idx0 + row_off][:, idx1 + col_off]) + w_add * add  (jamie.py:586-604) for a dense matrix."""
    _call('jamie_dense_block', ptr(Mat), Mat.stride(0), ptr(idx0), ptr(idx1), idx0.numel(), idx1.numel(), int(row_off),
          int(col_off), int(normalise), float(w_blk), ptr(add), float(w_add), ptr(out), _stream())


def axpby(out, a, x, b=0.0, y=None):
    _call('jamie_axpby', ptr(out), float(a), ptr(x), float(b), ptr(y), out.numel(), _stream())


def colsum(X, M, N, ld, out, nslab=1, slab_stride=0, accumulate=False):
    _call('jamie_colsum', ptr(X), M, N, ld, nslab, slab_stride, ptr(out), int(accumulate), _stream())


def pd_workspace(m, n):
    a, b = C.c_longlong(), C.c_longlong()
    _check(load().jamie_pd_workspace(m, n, C.byref(a), C.byref(b)))
    return a.value, b.value


def pd_step(state, iteration):
    _call('jamie_pd_step', C.pointer(state), int(iteration), _stream())


def pd_alpha(G2, F, partials, inv_trkk, alpha):
    _call('jamie_pd_alpha', ptr(G2), ptr(F), F.numel(), ptr(partials), partials.numel(), float(inv_trkk), ptr(alpha),
          _stream())


def standardise_columns(X):
    """Device `preclass(axis=0)`: X [N, d] fp32 / fp64 on the GPU -> (fp32 standardised [N, d], mean [d] f64, std [d] f64)."""
    if X.dtype not in (torch.float32, torch.float64) or X.dim() != 2 or not X.is_contiguous():
        raise JamieHipError('standardise_columns needs a contiguous 2-D fp32 / fp64 GPU tensor')
    N, d = X.shape
    R = int(max(1, min(256, (N + 2047) // 2048)))
    part = torch.empty(R * d, dtype=torch.float64, device=X.device)
    mean = torch.empty(d, dtype=torch.float64, device=X.device)
    sd = torch.empty(d, dtype=torch.float64, device=X.device)
    out = torch.empty(N, d, dtype=torch.float32, device=X.device)
    f64 = int(X.dtype == torch.float64)
    _call('jamie_col_stats', ptr(X), f64, N, d, d, ptr(part), R, ptr(mean), ptr(sd), _stream())
    _call('jamie_standardise', ptr(X), f64, N, d, d, ptr(mean), ptr(sd), ptr(out), _stream())
    return out, mean, sd


class SqRanges:
    """Host description of the gradient ranges jamie_grad_sqnorm_ranges covers (kept alive for recorded plans)."""

    def __init__(self, ranges):
        self.count = len(ranges)
        self.off = (C.c_longlong * self.count)(*[int(a) for a, _ in ranges])
        self.len = (C.c_longlong * self.count)(*[int(b) for _, b in ranges])
        self.blocks = load().jamie_sqnorm_range_blocks(self.len, self.count)


def colsum_problems(items, accumulate=False):
    """[(X [M, N] contiguous fp32, out [N])] -> (ctypes array, count, partial slots: ceil(N / 64) each)."""
    probs = []
    for X, out in items:
        p = ColsumProblem()
        p.X, p.out, p.M, p.N, p.ld, p.nslab, p.slab_stride, p.accumulate = ptr(X), ptr(out), X.shape[0], X.shape[1], X.shape[1], 1, 0, int(accumulate)
        probs.append(p)
    arr = (ColsumProblem * len(probs))(*probs)
    arr._keep = items
    return arr, len(probs), sum((X.shape[1] + 63) // 64 for X, _ in items)


def grad_sqnorm_ranges(g, ranges, partials, state, g16=None, fin=None, colsums=None):
    """`g16` (flat bf16, same layout as g): also receives the bf16 copy of every range.  `fin` (a LatentM with defer_final):
    one extra workgroup finalises the latent backward pass; `colsums` (from colsum_problems): further extra workgroups write
    column sums into g; `partials` has one more slot per extra workgroup."""
    if fin is not None:
        arr, cnt = (colsums[0], colsums[1]) if colsums is not None else (None, 0)
        _call('jamie_grad_sqnorm_ranges_fin', ptr(g), ptr(g16), ranges.off, ranges.len, ranges.count, ptr(partials),
              partials.numel(), ptr(state), C.pointer(fin), arr, cnt, _stream())
        return
    if g16 is not None:
        _call('jamie_grad_sqnorm_ranges_g16', ptr(g), ptr(g16), ranges.off, ranges.len, ranges.count, ptr(partials),
              partials.numel(), ptr(state), _stream())
        return
    _call('jamie_grad_sqnorm_ranges', ptr(g), ranges.off, ranges.len, ranges.count, ptr(partials), partials.numel(),
          ptr(state), _stream())


def hybrid_assemble(pairs, pidx, r0, r1, num_corr, true_ratio, rng, rng_stream, idx0, idx1):
    _call('jamie_hybrid_assemble', ptr(pairs), ptr(pidx), ptr(r0), ptr(r1), idx0.numel(), int(num_corr), float(true_ratio),
          ptr(rng), int(rng_stream), ptr(idx0), ptr(idx1), _stream())
