"""sella_amd — MI355X-native implementation of Sella's inner saddle-point linear-algebra loop.

Host code is plain Python over a ctypes C ABI (include/sella_hip.h); all O(n^2)/O(n^3) work
runs in hand-written HIP kernels for gfx950 (libsella_hip.so).  There is no CPU fallback.
"""
__version__ = '0.1.0'
