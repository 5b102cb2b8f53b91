"""The g++ twin of the C ABI (csrc_cpu/occ4d_twin.cpp -> libocc4d_cpu.so): explicit opt-in ONLY.

SURVEY.md 8(b) asks for every entry point of the minimum set "with a CPU twin compiled by g++ for config 1" (BASELINE
configs[0]: "runs without a GPU").  The twin answers the same symbols with plain host loops in the reference's as-written
op order.  It is NOT a fallback: nothing in the package loads it unless a caller says

    import occlusions4d_amd as pk
    pk.cpu_twin.enable()          # builds libocc4d_cpu.so with g++ if needed, swaps the library handle
    ... product modules on CPU tensors ...
    pk.cpu_twin.disable()

Without that call the product raises NativeLibraryError when libocc4d.so is missing and rejects CPU tensors (tests/
test_abi.py).  While the twin is loaded, CUDA tensors are rejected instead.  Used by tests/test_cpu_twin.py: BASELINE
configs[0] through the product modules, and the REFERENCE's own perform_inference driving them (container only).
"""
import contextlib
import os
import shutil
import subprocess

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc_cpu', 'occ4d_twin.cpp')
LIB = os.path.join(HERE, 'libocc4d_cpu.so')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
# -ffp-contract=off: the kNN / FPS distance expressions are pinned without FMA, only std::fma fuses
# -march=x86-64-v3 (AVX2 + FMA units for the explicit std::fma), not -march=native: the .so built in the build container
# travels to the GPU box with the tree
FLAGS = ['-O3', '-std=c++17', '-fopenmp', '-fPIC', '-shared', '-ffp-contract=off', '-march=x86-64-v3', '-I' + INCLUDE]


def build(force=False):
    deps = [SRC, os.path.join(INCLUDE, 'occ4d.h')]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        gxx = shutil.which('g++')
        if gxx is None:
            raise _lib.NativeLibraryError('g++ not found: the CPU twin cannot be built')
        subprocess.run([gxx] + FLAGS + [SRC, '-o', LIB], check=True)
    return LIB


def enable():
    """Build (if stale) and load the twin in place of libocc4d.so for this process."""
    return _lib.load_cpu_twin(build())


def disable():
    _lib.unload_cpu_twin()


def enabled():
    return _lib.is_twin()


@contextlib.contextmanager
def loaded():
    enable()
    try:
        yield
    finally:
        disable()
