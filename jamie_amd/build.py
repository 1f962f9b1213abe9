"""Build libjamie_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB = os.path.join(_HERE, 'libjamie_hip.so')


def library_path():
    return LIB


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def build_library(force=False, verbose=False):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(_HERE, '..', 'include', '*.h'))
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in deps):
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', LIB] + srcs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
