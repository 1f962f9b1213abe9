"""Bindings of the EXPERIMENTS build (libjamie_hip_exp.so = the product library + `-DJAMIE_EXPERIMENTS`): kernels that were
built, tested and measured slower than the product path inside the step, kept so that the measurements in profiles/ can be repeated.
Nothing in the product imports this module (tests/test_host_cpu.py asserts it).  Load the experiments library with
JAMIE_LIB=<path of libjamie_hip_exp.so> (jamie_amd.build.build_experiments() builds it).

  gemm_bf16_ring*   persistent loader / consumer ring GEMM for the backward products (round 4: 51-58 us per launch against 45 us
                    for one workgroup per tile; profiles/r04_persistent_ring_ablations.log, r04_probe_ring_depth_and_persistent_ring.log)
  gemm_bf16_bn      Linear forward + BatchNorm in one launch with an in-launch split-K hand-off (round 3: +38 / +120 us per step)
  gemm_bf16_skinny  register-fed skinny products (round 3: 11.3 against 9.8 us per launch)
"""
import ctypes as C

import torch

from . import _native as nv


def available():
    """The loaded library is an experiments build."""
    return all(hasattr(nv.load(), name) for name in nv.EXPERIMENT_EXPORTS)


def _require():
    if not available():
        raise nv.JamieHipError('this entry point exists in the experiments build only: jamie_amd.build.build_experiments(), then '
                               'JAMIE_LIB=jamie_amd/libjamie_hip_exp.so')


RING_MAX_ITEMS = 48


def gemm_bf16_ring_plan(problems, n_wg, max_items=RING_MAX_ITEMS):
    """Static tile lists of the persistent backward launch (jamie_gemm_bf16_ring_plan): an int32 device tensor [n_wg, max_items]
    ((problem << 24) | tile, -1 terminated), or None where the launch does not fit (more than `max_items` tiles per workgroup)."""
    import numpy as np
    _require()
    arr = (nv.GemmProblem * len(problems))(*problems)
    host = np.empty(n_wg * max_items, dtype=np.int32)
    rc = nv.load().jamie_gemm_bf16_ring_plan(arr, len(problems), n_wg, max_items, host.ctypes.data_as(C.c_void_p))
    if rc != 0:
        return None
    dev = problems[0]._keep[0].device
    return torch.from_numpy(host).to(dev)


def gemm_bf16_ring(problems, sched, n_wg, err, ranges=None, max_items=RING_MAX_ITEMS):
    """The backward products of one layer (dX on W as stored, dW on the activations as stored) as ONE persistent launch of
    `n_wg` workgroups (one per CU) that stream their tile lists `sched` through an LDS ring (jamie_gemm_bf16_ring); `err`: a
    zeroed uint32 device word that a broken hand-off would set; `ranges` as in gemm_bf16."""
    _require()
    arr = (nv.GemmProblem * len(problems))(*problems)
    if ranges is not None:
        g, g16, rg, partials, state, fin = ranges
        nv._call('jamie_gemm_bf16_ring', arr, len(problems), nv.ptr(sched), n_wg, max_items, nv.ptr(g), nv.ptr(g16), rg.off, rg.len, rg.count,
              nv.ptr(partials), partials.numel(), nv.ptr(state), C.pointer(fin) if fin is not None else None, nv.ptr(err), nv._stream())
        return
    nv._call('jamie_gemm_bf16_ring', arr, len(problems), nv.ptr(sched), n_wg, max_items, None, None, None, None, 0, None, 0, None, None,
          nv.ptr(err), nv._stream())


def gemm_bf16_bn(problems, bn_problems, cfg, p_drop, rng, tickets, mode=2, momentum=0.1, eps=1e-5, slope=0.01):
    """Linear forward + BatchNorm + LeakyReLU + dropout in one launch (jamie_gemm_bf16_bn): the workgroups of a column strip
    hand their split-K slabs to each other inside the launch; `tickets`: zeroed int32 device tensor (4 + 2 per 128-column strip)."""
    _require()
    arr = (nv.GemmProblem * len(problems))(*problems)
    barr = (nv.BnFwdProblem * len(bn_problems))(*bn_problems)
    nv._call('jamie_gemm_bf16_bn', arr, barr, len(problems), cfg, p_drop, momentum, eps, slope, nv.ptr(rng), nv.ptr(tickets),
          tickets.numel(), int(mode), nv._stream())


def gemm_bf16_skinny(problems):
    """C[M, N <= 128] = A B^T on K-contiguous bf16 operands, fp32 written once (jamie_gemm_bf16_skinny)."""
    _require()
    arr = (nv.GemmProblem * len(problems))(*problems)
    nv._call('jamie_gemm_bf16_skinny', arr, len(problems), nv._stream())


