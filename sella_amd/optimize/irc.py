"""Intrinsic reaction coordinate — contract of sella/optimize/irc.py:20-175 (`IRC(atoms, ...).run(fmax, fmax_inner,
steps, direction)`), rebuilt around the device step solver.

The path of steepest descent in mass-weighted coordinates is followed from a first-order saddle in arcs of
mass-weighted length `dx`.  Every arc is a constrained minimisation: find the point on the sphere of radius dx
around the arc's pivot (the previous path point) where the mass-weighted gradient is parallel to the displacement
from the pivot.  That inner problem is exactly a restricted step — family `QuasiNewtonIRC`, measure `sphere`
(`sella_restricted_step`, csrc/stepper.hip: |(s + d1) * sqrt(m)| = dx, restricted_step.py:145-158,
stepper.py:99-111) — re-solved after every force call until the tangential force on the sphere is below
`fmax_inner`.

State kept by the driver: the saddle (geometry, PES caches, Hessian) so that the second direction starts from it,
the transition vector scaled to the first arc, and `pivot_offset`, the displacement accumulated within the current
arc (the reference's `d1`).
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


class _Saddle:
    """What has to be put back to follow the path in the other direction (irc.py:95-106)."""

    def __init__(self, pes):
        self.x = pes.get_x().copy()
        self.hessian = pes.get_H().asarray().copy()
        self.curr, self.last = pes.curr.copy(), pes.last.copy()

    def restore(self, pes):
        pes.set_x(self.x)
        pes.curr, pes.last = self.curr.copy(), self.last.copy()
        pes.set_H(self.hessian.copy(), initialized=True)


class IRC(Optimizer):
    def __init__(self, atoms, logfile='-', trajectory=None, master=None, ninner_iter=10, irctol=1e-2, dx=0.1,
                 eta=1e-4, gamma=0.1, peskwargs=None, keep_going=False, **kwargs):
        Optimizer.__init__(self, atoms, restart=None, logfile=logfile, trajectory=trajectory, master=master)
        self.ninner_iter, self.irctol, self.dx, self.keep_going = ninner_iter, irctol, dx, keep_going
        self.peskwargs = {'gamma': gamma} if peskwargs is None else peskwargs
        self.sqrtm = np.sqrt(np.repeat(self.atoms.get_masses(), 3))
        self.pes = PES(atoms, eta=eta, proj_trans=False, proj_rot=False, **kwargs)
        self.saddle = None
        self.v0ts = None                 # transition vector, Cartesian, mass-weighted length dx
        self.pivot_offset = None
        self.first = True

    # attribute names of the reference driver, for scripts that read them
    @property
    def d1(self):
        return self.pivot_offset

    @property
    def x0(self):
        return None if self.saddle is None else self.saddle.x

    @property
    def H0(self):
        return None if self.saddle is None else self.saddle.hessian

    def _characterise_saddle(self):
        """Diagonalise at the saddle and take the lowest mode of the MASS-WEIGHTED Hessian as the initial tangent
        (irc.py:85-100); its sign is fixed by making the first non-zero component positive."""
        self.pes.kick(0, True, **self.peskwargs)
        self.saddle = _Saddle(self.pes)
        weighted = self.saddle.hessian / (self.sqrtm[:, None] * self.sqrtm[None, :])
        mode = eigh(weighted, subset_by_index=[0, 0])[1][:, 0] / self.sqrtm
        lead = mode[np.flatnonzero(mode)[0]]
        self.v0ts = self.dx * mode * (1.0 if lead > 0 else -1.0)

    def irun(self, fmax=0.05, fmax_inner=0.01, steps=None, direction='forward'):
        sign = {'forward': 1.0, 'reverse': -1.0}.get(direction)
        if sign is None:
            raise ValueError('direction must be one of "forward" or "reverse"!')
        if self.saddle is None:
            self._characterise_saddle()
        else:
            self.saddle.restore(self.pes)
        self.pivot_offset = sign * self.v0ts
        self.first = True
        self.fmax_inner = min(fmax, fmax_inner)
        return Optimizer.irun(self, fmax, 100000000 if steps is None else steps)

    def run(self, *args, **kwargs):
        done = False
        for done in self.irun(*args, **kwargs):
            pass
        return done

    def _tangential_fmax(self):
        """Largest per-atom force on the sphere: the mass-weighted gradient with its component along the
        mass-weighted displacement from the pivot removed, mapped back to Cartesian forces (irc.py:129-137)."""
        radial = self.pivot_offset * self.sqrtm
        radial /= np.linalg.norm(radial)
        gw = self.pes.get_g() / self.sqrtm
        tangential = (gw - radial * (radial @ gw)) * self.sqrtm
        return float(np.sqrt((tangential.reshape((-1, 3)) ** 2).sum(axis=1).max()))

    def step(self):
        if self.first:
            # first arc: straight along the transition vector
            self.pes.kick(self.pivot_offset)
            self.first = False
        for _ in range(self.ninner_iter):
            move, size = IRCTrustRegion(self.pes, 0, self.dx, method=QuasiNewtonIRC, sqrtm=self.sqrtm,
                                        d1=self.pivot_offset, W=self.get_W()).get_s()
            on_sphere = abs(size - self.dx) < 1e-8
            self.pivot_offset = self.pivot_offset + move
            self.pes.kick(move)
            if (on_sphere and self._tangential_fmax() < self.fmax_inner) or self.converged():
                break
        else:
            if not self.keep_going:
                raise IRCInnerLoopConvergenceFailure
            warnings.warn('IRC inner loop failed to converge! The trajectory is no longer a trustworthy IRC.')
        self.pivot_offset = np.zeros_like(self.pivot_offset)          # the point reached is the next pivot

    def converged(self, forces=None):
        """A minimum: projected forces below fmax AND no negative curvature left (irc.py:152-156)."""
        if self.first:
            return False
        lowest = self.pes.H.evals
        return bool(self.pes.converged(self.fmax)[0]) and lowest is not None and bool(lowest[0] > 0)

    def log(self, forces=None):
        if self.logfile is None:
            return
        fmax = self.pes.converged(self.fmax)[1]
        if self.nsteps == 0:
            self.logfile.write('%s %4s %15s %12s\n' % (' ' * 3, 'Step', 'Energy', 'fmax'))
        self.logfile.write('IRC %4d %15.6f %12.4f\n' % (self.nsteps, self.pes.get_f(), fmax))
        self.logfile.flush()

    def get_W(self):
        """Diagonal of irc.py:174-175's mass weighting W = diag(1 / sqrt(m)): a row scaling, never an n x n matrix."""
        return 1. / self.sqrtm
