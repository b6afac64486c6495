"""sella_amd — MI355X-native implementation of Sella's inner saddle-point linear-algebra loop.

Host code is plain Python over a ctypes C ABI (include/sella_hip.h); all O(n^2)/O(n^3) work
runs in hand-written HIP kernels for gfx950 (libsella_hip.so).  There is no CPU fallback.

    from sella_amd import Sella, Constraints        # same entry points as `from sella import ...`
"""
__version__ = '0.1.0'

from .utilities.hostcpu import limit_blas_threads as _limit_blas_threads  # noqa: E402

_limit_blas_threads()          # see utilities/hostcpu.py: BLAS pools capped at the CPUs this process may use

_LAZY = {
    'Sella': ('sella_amd.optimize.optimize', 'Sella'),
    'IRC': ('sella_amd.optimize.irc', 'IRC'),
    'PES': ('sella_amd.peswrapper', 'PES'),
    'InternalPES': ('sella_amd.peswrapper', 'InternalPES'),
    'InternalCoordinates': ('sella_amd.internal', 'InternalCoordinates'),
    'Constraints': ('sella_amd.internal', 'Constraints'),
    'Atoms': ('sella_amd.atoms', 'Atoms'),
    'LibrarySearch': ('sella_amd.search', 'LibrarySearch'),
    'EnsembleThreads': ('sella_amd.ensemble', 'EnsembleThreads'),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod), attr)
    raise AttributeError(name)
