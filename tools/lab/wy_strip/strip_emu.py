import sys, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/hostemu')
import numpy as np
import build_emu
from sella_amd import _lib
_lib._set_library_for_tests(ctypes.CDLL(build_emu.build()))
from sella_amd.device import Context
ctx = Context(0)
ctx.set_option('eigh_wy_nb64_min', 1)
for n in (64, 128, 192, 320):
    rng = np.random.RandomState(n)
    A = rng.normal(size=(n, n)); A = A + A.T
    out = []
    for strip in (0, 2):
        ctx.set_option('eigh_wy_strip', strip)
        w, V, Vt = ctx.eigh(ctx.upload(A))
        out.append((np.array(w), V.numpy()))
    Vn = out[1][1]
    print(n, 'ld==n?', 'max |dV|', np.abs(out[0][1] - out[1][1]).max(), 'resid', np.abs(A @ Vn - Vn * out[1][0]).max(), 'orth', np.abs(Vn.T @ Vn - np.eye(n)).max())
