#!/usr/bin/env python3
"""BASELINE configs[2] in synthetic form: 1024-atom fcc(111) Cu slab, bonds = nearest
neighbours, angles from bond pairs, random tangent.  Times q + B-matrix rows + D(v) on the device kernel
(`sella_internals_eval`) and the host-side assembly of the dense matrices around it."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sella_amd import device as _dev  # noqa: E402
from sella_amd.atoms import fcc111  # noqa: E402
from sella_amd.device import Context  # noqa: E402
from sella_amd.internal import InternalCoordinates, angles_from_bonds, neighbour_bonds  # noqa: E402

size = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 8, 16)
ctx = Context()
_dev._default = ctx
slab = fcc111('Cu', size, vacuum=7.5)
rng = np.random.RandomState(0)
slab.positions += 0.03 * rng.normal(size=slab.positions.shape)
t0 = time.perf_counter()
bonds, bncv = neighbour_bonds(slab, 1.25 * 3.61 / np.sqrt(2))
angles, ancv = angles_from_bonds(bonds, bncv)
t_topo = time.perf_counter() - t0
ic = InternalCoordinates(slab, bonds=bonds, angles=angles, bond_ncvecs=bncv, angle_ncvecs=ancv)
v = rng.normal(size=ic.ndof)
for name in ('bonds', 'angles'):
    pos, tvec, dofs = ic._batch(name)
    tan = v[dofs].reshape(pos.shape)
    ctx.internals_eval(pos, tvec, tan)
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ctx.internals_eval(pos, tvec, tan)
    dt = (time.perf_counter() - t0) / reps
    ctx.prof_enable(False)
    p = ctx.prof_get(3)
    us = 1e3 * p['ms'] / max(1, p['launches'])
    print(json.dumps(dict(op='q + dq/dx + H.t', kind=name, coordinates=len(pos), kernel_us=round(us, 2),
                          kernel_Mcoord_per_s=round(len(pos) / us, 1), kernel_GBps=round(p['bytes'] / max(1, p['launches']) / us / 1e3, 1),
                          api_call_us=round(1e6 * dt, 1))), flush=True)
t0 = time.perf_counter()
B = ic.jacobian()
t_B = time.perf_counter() - t0
t0 = time.perf_counter()
D = ic.hessian_rdot(v)
t_D = time.perf_counter() - t0
print(json.dumps(dict(atoms=len(slab), nint=ic.nint, ndof=ic.ndof, topology_s=round(t_topo, 3),
                      dense_B_assembly_ms=round(1e3 * t_B, 1), dense_D_assembly_ms=round(1e3 * t_D, 1),
                      B_shape=list(B.shape))), flush=True)
