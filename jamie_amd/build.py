"""Build libjamie_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every `csrc/*.hip` is compiled to its own object under `csrc/_obj/` (in parallel, only when the source or a header
is newer) and the objects are linked into `libjamie_hip.so`."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(_HERE, 'libjamie_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
# Per-source flags.  gemm_f32.hip: the kernels index their by-value argument struct with a run-time problem number; clang copies a
# by-value aggregate into a private alloca and InstCombine forwards the loads back to the kernarg segment only while the alloca has
# at most 300 users -- the 256 x 128 bf16x3 instantiation has more, kept the 2.5 KB copy in SCRATCH, every descriptor field in a
# vector register and a readfirstlane loop in front of every buffer load (twice the run time).
FILE_FLAGS = {'gemm_f32.hip': ['-mllvm', '-instcombine-max-copied-from-constant-users=100000']}


def file_flags(src):
    return FILE_FLAGS.get(os.path.basename(src), [])


def library_path():
    return LIB


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else -1.0


def build_variant(tag, flags, verbose=False, only=None):
    """An A/B build of the same sources with extra hipcc flags (e.g. -DJAMIE_OLD_REDUCE): `libjamie_hip_<tag>.so` next to
    the product library, selected at run time with JAMIE_LIB=<path> (tools/ab.sh).  `only`: the source files (base names) the
    flags concern; the other objects are the product build's."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    odir = os.path.join(CSRC, '_obj_' + tag)
    os.makedirs(odir, exist_ok=True)
    srcs = sources()
    if only is not None:
        build_library()
    objs = [os.path.join(odir if (only is None or os.path.basename(s) in only) else OBJ, os.path.basename(s)[:-4] + '.o')
            for s in srcs]

    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(_HERE, '..', 'include', '*.h'))
    hnew = max([_mtime(h) for h in headers] + [_mtime(__file__)])
    stamp = os.path.join(odir, 'FLAGS')
    fresh = os.path.exists(stamp) and open(stamp).read() == ' '.join(flags)

    def cc(so):
        if (only is None or os.path.basename(so[0]) in only) and not (fresh and _mtime(so[1]) >= max(_mtime(so[0]), hnew)):
            subprocess.run([hipcc] + FLAGS + file_flags(so[0]) + list(flags) + ['-c', so[0], '-o', so[1]], check=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    with open(stamp, 'w') as f:
        f.write(' '.join(flags))
    out = os.path.join(_HERE, f'libjamie_hip_{tag}.so')
    if _mtime(out) < max(_mtime(o) for o in objs):
        subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, check=True)
    if verbose:
        print('built', out)
    return out


EXPERIMENT_SOURCES = ['gemm_bf16.hip', 'gemm_bf16_ring.hip', 'skinny.hip', 'bn_act.hip']


def build_experiments(verbose=False):
    """libjamie_hip_exp.so: the product library + the kernels behind `-DJAMIE_EXPERIMENTS` (built, tested, measured slower: see
    jamie_amd/experiments.py).  Selected with JAMIE_LIB=<its path>; tests/experiments/ and the probe tools use it."""
    return build_variant('exp', ['-DJAMIE_EXPERIMENTS'], verbose=verbose, only=EXPERIMENT_SOURCES)


def build_library(force=False, verbose=False, jobs=None):
    srcs = sources()
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(_HERE, '..', 'include', '*.h'))
    hnew = max([_mtime(h) for h in headers] + [_mtime(__file__)])
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('JAMIE_HIPCC_FLAGS', '').split()
    os.makedirs(OBJ, exist_ok=True)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in srcs]
    # the flag string the objects were built with is recorded: a build with other flags (a diagnostic JAMIE_HIPCC_FLAGS build, or
    # the plain build after one) recompiles everything instead of trusting the time stamps -- otherwise a later plain build would
    # keep serving the instrumented library
    stamp = os.path.join(OBJ, 'FLAGS')
    flags_now = ' '.join(FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _mtime(o) < max(_mtime(s), hnew)]
    if not todo and _mtime(LIB) >= max(_mtime(o) for o in objs):
        return LIB

    def cc(so):
        cmd = [hipcc] + FLAGS + file_flags(so[0]) + extra + ['-c', so[0], '-o', so[1]]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(cc, todo))
    with open(stamp, 'w') as f:
        f.write(flags_now)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB
