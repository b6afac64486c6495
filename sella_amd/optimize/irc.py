"""Intrinsic reaction coordinate driver — drop-in for sella/optimize/irc.py:20-175.

From a first-order saddle the path of steepest descent in mass-weighted coordinates is followed in
steps of length `dx`: each outer step solves, in an inner loop, for the point on the mass-weighted sphere
of radius dx around the previous path point where the gradient is parallel to the displacement
(`IRCTrustRegion` + `QuasiNewtonIRC`, restricted_step.py:145-158, stepper.py:99-111).  Same constructor
keywords, same `run(fmax, fmax_inner, steps, direction)`; the linear algebra goes through the device
classes of this package.  Atomic masses come from `atoms.get_masses()`.
"""
import warnings

import numpy as np
from scipy.linalg import eigh

from ..atoms import Optimizer
from ..peswrapper import PES
from .restricted_step import IRCTrustRegion
from .stepper import QuasiNewtonIRC


class IRCInnerLoopConvergenceFailure(RuntimeError):
    pass


class IRC(Optimizer):
    def __init__(self, atoms, logfile='-', trajectory=None, master=None, ninner_iter=10, irctol=1e-2, dx=0.1,
                 eta=1e-4, gamma=0.1, peskwargs=None, keep_going=False, **kwargs):
        Optimizer.__init__(self, atoms, restart=None, logfile=logfile, trajectory=None, master=master)
        self.ninner_iter = ninner_iter
        self.irctol = irctol
        self.dx = dx
        self.peskwargs = dict(gamma=gamma) if peskwargs is None else peskwargs
        self.sqrtm = np.repeat(np.sqrt(self.atoms.get_masses()), 3)
        self.pes = PES(atoms, eta=eta, proj_trans=False, proj_rot=False, **kwargs)
        self.lastrun = None
        self.x0 = self.pes.get_x().copy()
        self.v0ts = None
        self.H0 = None
        self.peslast = None
        self.xi = 1.
        self.first = True
        self.keep_going = keep_going

    def irun(self, fmax=0.05, fmax_inner=0.01, steps=None, direction='forward'):
        if direction not in ['forward', 'reverse']:
            raise ValueError('direction must be one of "forward" or "reverse"!')
        if self.v0ts is None:
            # initial diagonalisation: the transition vector in mass-weighted coordinates (:85-100)
            self.pes.kick(0, True, **self.peskwargs)
            self.H0 = self.pes.get_H().asarray().copy()
            Hw = self.H0 / np.outer(self.sqrtm, self.sqrtm)
            _, vecs = eigh(Hw)
            self.v0ts = self.dx * vecs[:, 0] / self.sqrtm
            if self.v0ts[np.nonzero(self.v0ts)[0][0]] < 0:
                self.v0ts *= -1
            self.pescurr = self.pes.curr.copy()
            self.peslast = self.pes.last.copy()
        else:
            # restore the saddle for the other direction (:101-106)
            self.pes.set_x(self.x0)
            self.pes.curr = self.pescurr.copy()
            self.pes.last = self.peslast.copy()
            self.pes.set_H(self.H0.copy(), initialized=True)
        self.d1 = self.v0ts.copy() if direction == 'forward' else -self.v0ts.copy()
        self.first = True
        self.fmax_inner = min(fmax, fmax_inner)
        return Optimizer.irun(self, fmax, steps if steps is not None else 100000000)

    def run(self, *args, **kwargs):
        converged = False
        for converged in self.irun(*args, **kwargs):
            pass
        return converged

    def step(self):
        if self.first:
            self.pes.kick(self.d1)
            self.first = False
        for _ in range(self.ninner_iter):
            s, smag = IRCTrustRegion(self.pes, 0, self.dx, method=QuasiNewtonIRC, sqrtm=self.sqrtm, d1=self.d1,
                                     W=self.get_W()).get_s()
            bound_clip = abs(smag - self.dx) < 1e-8
            self.d1 += s
            self.pes.kick(s)
            g1 = self.pes.get_g()
            d1m = self.d1 * self.sqrtm
            d1m /= np.linalg.norm(d1m)
            g1m = g1 / self.sqrtm
            g1m_proj = g1m - d1m * (d1m @ g1m)
            fmax = np.linalg.norm((g1m_proj * self.sqrtm).reshape((-1, 3)), axis=1).max()
            if bound_clip and fmax < self.fmax_inner:
                break
            elif self.converged():
                break
        else:
            if self.keep_going:
                warnings.warn('IRC inner loop failed to converge! The trajectory is no longer a trustworthy IRC.')
            else:
                raise IRCInnerLoopConvergenceFailure
        self.d1 *= 0.

    def converged(self, forces=None):
        if self.first:
            return False
        evals = self.pes.H.evals
        return bool(self.pes.converged(self.fmax)[0] and evals is not None and evals[0] > 0)

    def log(self, forces=None):
        if self.logfile is None:
            return
        _, fmax, _ = self.pes.converged(self.fmax)
        e = self.pes.get_f()
        if self.nsteps == 0:
            self.logfile.write('%s %4s %15s %12s\n' % (' ' * 3, 'Step', 'Energy', 'fmax'))
        self.logfile.write('IRC %4d %15.6f %12.4f\n' % (self.nsteps, e, fmax))
        self.logfile.flush()

    def get_W(self):
        return 1. / self.sqrtm          # diagonal of irc.py:174-175's W (a row scaling, never formed as n x n)
