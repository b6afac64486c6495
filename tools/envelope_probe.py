#!/usr/bin/env python3
"""Device Davidson trajectory against the reference's digest and its own sensitivity envelope (tests/golden/big_digests.json):
|lam0_device(j) - lam0_reference(j)| / envelope(j) for every vector count j.  Usage: tools/envelope_probe.py [n ...]"""
import json
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import hessian_like  # noqa: E402
from sella_amd.device import Context  # noqa: E402

dig = json.load(open(os.path.join(R, 'tests', 'golden', 'big_digests.json')))
ctx = Context(0)
for n in [int(a) for a in sys.argv[1:]] or [300, 768, 3072]:
    d = dig[str(n)]
    env = d['envelope']['max_abs_change']
    A, P, g = hessian_like(n, 0, eps=5e-3)
    dA, dP = ctx.upload(A), ctx.upload(P)
    w, Q, Qt = ctx.eigh(dP)
    out = []
    for j in range(2, len(env) + 1):
        lams, V, AV, nmv = ctx.davidson(dA, n, g, 0.1, method='jd0', maxiter=j, Pvecs=Q, PvecsT=Qt, pevals=w)
        if V.shape[1] < j:
            out.append(f'(exit at {V.shape[1]})')
            break
        dev = abs(lams[0] - d['ritz'][j - 1])
        out.append(f'{j}:{dev:.1e}/{env[j - 1]:.1e}={dev / max(env[j - 1], 1e-300):.2g}')
    print(f'n={n}:', ' '.join(out), flush=True)
