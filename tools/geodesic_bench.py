#!/usr/bin/env python3
"""BASELINE configs[2]: 1024-atom Cu(111) slab in redundant internal coordinates (nearest-neighbour bonds,
periodic in x and y), EMT on the device: spectral factor of the B-matrix, geodesic steps, one full
`InternalPES.kick` and (optionally) `Sella(internal=...)` steps.

    python tools/geodesic_bench.py [nx ny nz] [--steps K] [--sella-steps S] [--angles]

`SELLA_EMU=1` runs the same script on the host emulation of the kernels (small sizes only; a dry run)."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get('SELLA_EMU'):
    sys.path.insert(0, os.path.join(REPO, 'tests', 'hostemu'))
    import build_emu  # noqa: E402
    from sella_amd import _lib  # noqa: E402
    _lib._set_library_for_tests(ctypes.CDLL(build_emu.build()))

from sella_amd import device as _dev  # noqa: E402
from sella_amd.atoms import EMT, fcc111  # noqa: E402
from sella_amd.device import Context  # noqa: E402
from sella_amd.internal import InternalCoordinates, angles_from_bonds, neighbour_bonds  # noqa: E402
from sella_amd.peswrapper import InternalPES, _BFactor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('size', nargs='*', type=int, default=[8, 8, 16])
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--sella-steps', type=int, default=0)
ap.add_argument('--angles', action='store_true', help='also time the factor with all bond angles added')
ap.add_argument('--dx', type=float, default=0.02)
ap.add_argument('--profile', action='store_true', help='cProfile the Sella steps')
args = ap.parse_args()

ctx = Context()
_dev._default = ctx


def out(**kw):
    print(json.dumps(kw), flush=True)


def clock(f, *a, **k):
    t0 = time.perf_counter()
    r = f(*a, **k)
    return r, time.perf_counter() - t0


slab = fcc111('Cu', tuple(args.size), vacuum=7.5)
rng = np.random.RandomState(0)
slab.positions += 0.03 * rng.normal(size=slab.positions.shape)
slab.calc = EMT()
(bonds, bncv), t_topo = clock(neighbour_bonds, slab, 1.25 * 3.61 / np.sqrt(2))
ic = InternalCoordinates(slab, bonds=bonds, bond_ncvecs=bncv)
out(atoms=len(slab), ndof=ic.ndof, bonds=len(bonds), topology_s=round(t_topo, 3))

if args.angles:
    angles, ancv = angles_from_bonds(bonds, bncv)
    ica = InternalCoordinates(slab, bonds=bonds, angles=angles, bond_ncvecs=bncv, angle_ncvecs=ancv)
    Bs, t_B = clock(ica.jacobian_csr)
    _BFactor(Bs)
    fac, t_f = clock(_BFactor, Bs)
    y = rng.normal(size=(ica.nint, 2))
    fac.pinv_dot(y)
    _, t_p = clock(fac.pinv_dot, y)
    out(case='bonds + angles', nint=ica.nint, rank=fac.rank, smin=float(fac.s[-1]), smax=float(fac.s[0]),
        sparse_B_ms=round(1e3 * t_B, 1), factor_ms=round(1e3 * t_f, 1), pinv_dot_2rhs_ms=round(1e3 * t_p, 2))
    del fac, ica

# ---- factor + PES construction ------------------------------------------------------------------------
Bs, t_B = clock(ic.jacobian_csr)
_BFactor(Bs)                                           # warm-up (allocations, first eigh of this size)
fac, t_f = clock(_BFactor, Bs)
# B B^+ B = B on random probes
probe = rng.normal(size=(ic.ndof, 3))
BP = Bs @ probe
err = np.abs(Bs @ fac.pinv_dot(BP) - BP).max() / np.abs(BP).max()
out(case='bonds', nint=ic.nint, rank=fac.rank, smin=float(fac.s[-1]), smax=float(fac.s[0]),
    sparse_B_ms=round(1e3 * t_B, 1), factor_ms=round(1e3 * t_f, 1), B_Bpinv_B_rel_err=float(err))
del fac

pes, t_init = clock(InternalPES, slab, ic)
g, t_g = clock(pes.get_g)
out(stage='InternalPES', dim=pes.dim, construct_s=round(t_init, 3), first_gradient_s=round(t_g, 3),
    g_int_norm=float(np.linalg.norm(g)))

# ---- geodesic steps to feasible targets -----------------------------------------------------------------
x0 = slab.positions.copy()
q0 = pes.get_x()
times, errs, nfev = [], [], []
for k in range(args.steps):
    slab.positions = x0 + args.dx * rng.normal(size=x0.shape)
    q1 = pes.int.calc()
    slab.positions = x0.copy()
    pes.get_g()
    calls0 = ctx_calls = None
    t0 = time.perf_counter()
    dx_i, dx_f, g_par = pes.set_x(q1)
    times.append(time.perf_counter() - t0)
    q = pes.int.calc()
    errs.append(float(np.abs(q - q1).max() / np.abs(q1 - q0).max()))
slab.positions = x0.copy()
# the reference's default: pseudo-inverse re-evaluated at every point of the path (here: carried by PCG)
pes_x = InternalPES(slab, ic, exact_geodesic=True)
pes_x.get_g()
tx, ex = [], []
for k in range(args.steps):
    slab.positions = x0 + args.dx * rng.normal(size=x0.shape)
    q1 = pes_x.int.calc()
    slab.positions = x0.copy()
    pes_x.get_g()
    t0 = time.perf_counter()
    pes_x.set_x(q1)
    tx.append(time.perf_counter() - t0)
    ex.append(float(np.abs(pes_x.int.calc() - q1).max() / np.abs(q1 - q0).max()))
slab.positions = x0.copy()
out(stage='EXACT geodesic step (set_x, exact_geodesic=True)', steps=args.steps,
    ms_per_step=round(1e3 * float(np.median(tx)), 2), ms_all=[round(1e3 * t, 1) for t in tx], rel_target_error_max=max(ex))
del pes_x
out(stage='geodesic step (set_x)', steps=args.steps, cart_rms_displacement=args.dx,
    ms_per_step=round(1e3 * float(np.median(times)), 2), ms_all=[round(1e3 * t, 1) for t in times],
    rel_target_error_max=max(errs))

# ---- one full kick: geodesic + force call + quasi-Newton update of the nint x nint Hessian ----------------
pes.get_g()
dq = 0.3 * (pes.int.calc() * 0 + (Bs @ (args.dx * rng.normal(size=ic.ndof))))
ratio, t_k1 = clock(pes.kick, dq)
ratio2, t_k2 = clock(pes.kick, -0.5 * dq)
out(stage='kick', first_s=round(t_k1, 3), second_s=round(t_k2, 3), ratio=None if ratio is None else float(ratio),
    ratio2=None if ratio2 is None else float(ratio2))

if args.sella_steps:
    from sella_amd import Sella
    slab.positions = x0.copy()
    # (exact_geodesic=False: the pseudo-inverse of the starting point along the path; the reference default
    # re-factorises B at every right-hand side, 25 x 130 ms per step at this size)
    dyn = Sella(slab, internal=ic, logfile='-', order=0, exact_geodesic=bool(os.environ.get('EXACT_GEODESIC')))
    _, t_s1 = clock(dyn.run, 1e-3, 1)
    n0 = slab.calc.ncalls
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    _, t_s = clock(dyn.run, 1e-3, args.sella_steps)
    if args.profile:
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
        pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
    out(stage='Sella(internal)', first_step_s=round(t_s1, 2), steps=args.sella_steps,
        s_per_step=round(t_s / args.sella_steps, 3), force_calls=slab.calc.ncalls - n0)
