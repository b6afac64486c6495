#!/usr/bin/env python3
"""Which option changes the eigendecomposition at 3N = 3072 bit for bit (default configuration)?"""
import os, sys
sys.path.insert(0, os.path.abspath(os.environ['SELLA_AB_ROOT']) if os.environ.get('SELLA_AB_ROOT') else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sella_amd.device import Context
ctx = Context(0)
for kv in filter(None, os.environ.get('EIGH_OPTS', '').split(',')):
    key, value = kv.split('=')
    ctx.set_option(key, int(value))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
rng = np.random.RandomState(0)
A = rng.normal(size=(n, n)); A = A + A.T
dA = ctx.upload(A)
def run():
    w, V, Vt = ctx.eigh(dA)
    out = (np.array(w), V.numpy())
    V.free(); Vt.free()
    return out
ref = run()
again = run()
print('run to run identical:', np.array_equal(ref[0], again[0]) and np.array_equal(ref[1], again[1]))
import hashlib
print('digest', hashlib.sha256(ref[0].tobytes() + ref[1].tobytes()).hexdigest()[:16], 'w only', hashlib.sha256(ref[0].tobytes()).hexdigest()[:16])
np.save(os.environ.get('BITCHECK_OUT', '/tmp/bitcheck_w.npy'), ref[0])
for key, val, back in (('eigh_gemv_flat', 0, 1), ('eigh_dc_pipeline', 0, 1), ('rank2k_fixed', 0, 1), ('eigh_wy_overlap', 0, 1)):
    ctx.set_option(key, val)
    o = run()
    ctx.set_option(key, back)
    print(f'{key}={val}: w identical {np.array_equal(ref[0], o[0])}, V identical {np.array_equal(ref[1], o[1])}, max |dw| {np.abs(ref[0] - o[0]).max():.2e}')
