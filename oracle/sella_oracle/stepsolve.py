"""ORACLE (test infrastructure, never imported by the product).

Step solve of the saddle-point loop restated from
sella/optimize/stepper.py:20-199 (QN / RFO / P-RFO step families) and
sella/optimize/restricted_step.py:11-253 (1-D root find on the step-length
parameter alpha under a trust / per-atom / per-internal constraint).
"""
import numpy as np
from scipy.linalg import eigh


# ---------------------------------------------------------------- steppers --
class NaiveStep:                                         # stepper.py:44-55
    alpha0, alphamin, alphamax, slope, newton_safe = 0.5, 0.0, 1.0, 1.0, True

    def __init__(self, dx):
        self.dx = dx

    def get_s(self, alpha):
        return alpha * self.dx, self.dx


class QuasiNewtonStep:                                   # stepper.py:58-96
    alpha0, alphamin, alphamax, slope, newton_safe = 0.0, 0.0, np.inf, -1, True

    def __init__(self, g, H, order=0, d1=None):
        self.g, self.H, self.order, self.d1 = g, H, order, d1
        evals, evecs = H.evals, H.evecs
        if evals is None:                                # stepper.py:62-64
            evals, evecs = eigh(H.asarray())
        self.L = np.abs(evals)
        self.L[:order] *= -1
        self.V = evecs
        self.Vg = self.V.T @ g
        self.ones = np.ones_like(self.L)
        self.ones[:order] = -1

    def get_s(self, alpha):
        den = self.L + alpha * self.ones
        sp = self.Vg / den
        return -self.V @ sp, self.V @ (sp / den)


class QuasiNewtonIRCStep(QuasiNewtonStep):               # stepper.py:99-111
    """Quasi-Newton step towards the IRC constraint sphere centred at -d1."""

    def __init__(self, g, H, order=0, d1=None):
        QuasiNewtonStep.__init__(self, g, H, order, d1)
        self.Vd1 = self.V.T @ d1

    def get_s(self, alpha):
        den = np.abs(self.L) + alpha
        sp = -(self.Vg + alpha * self.Vd1) / den
        return self.V @ sp, -self.V @ ((sp + self.Vd1) / den)


class RFOStep:                                           # stepper.py:114-157
    alpha0, alphamin, alphamax, slope, newton_safe = 1.0, 0.0, 1.0, 1.0, False

    def __init__(self, g, H, order=0, d1=None):
        self.g, self.H, self.order, self.d1 = g, H, order, d1
        self.A = np.block([[H.asarray(), g[:, None]], [g, 0]])

    def get_s(self, alpha):
        o = self.order
        A = self.A * alpha
        A[:-1, :-1] *= alpha
        L, V = eigh(A)
        den = V[-1, o]
        if abs(den) < 1e-12:
            den = np.sign(den) * 1e-12 if den != 0 else 1e-12
        s = V[:-1, o] * alpha / den

        dA = self.A.copy()
        dA[:-1, :-1] *= 2 * alpha
        V1 = np.delete(V, o, 1)
        gap = np.delete(L, o) - L[o]
        gap = np.where(gap >= 0, np.maximum(gap, 1e-12), np.minimum(gap, -1e-12))
        dV = V1 @ ((V1.T @ (dA @ V[:, o])) / gap)
        dsda = (V[:-1, o] / den + (alpha / den) * dV[:-1]
                - (V[:-1, o] * alpha / den ** 2) * dV[-1])
        return s, dsda


class PRFOStep:                                          # stepper.py:160-185
    alpha0, alphamin, alphamax, slope, newton_safe = 1.0, 0.0, 1.0, 1.0, False

    def __init__(self, g, H, order=0, d1=None):
        self.g, self.H, self.order, self.d1 = g, H, order, d1
        self.Vmax = H.evecs[:, :order]
        self.Vmin = H.evecs[:, order:]
        self.max = RFOStep(self.Vmax.T @ g, H.project(self.Vmax),
                           order=self.Vmax.shape[1])
        self.min = RFOStep(self.Vmin.T @ g, H.project(self.Vmin), order=0)

    def get_s(self, alpha):
        smax, dmax = self.max.get_s(alpha)
        smin, dmin = self.min.get_s(alpha)
        return (self.Vmax @ smax + self.Vmin @ smin,
                self.Vmax @ dmax + self.Vmin @ dmin)


_STEPPERS = {
    QuasiNewtonStep: ['qn', 'quasi-newton', 'quasi newton', 'newton', 'mmf',
                      'minimum mode following', 'minimum-mode following',
                      'dimer'],
    RFOStep: ['rfo', 'rational function optimization'],
    PRFOStep: ['prfo', 'p-rfo', 'partitioned rational function optimization'],
}


def get_stepper(name):                                   # stepper.py:195-199
    for cls, names in _STEPPERS.items():
        if name in names:
            return cls
    raise ValueError("Unknown stepper name: {}".format(name))


# --------------------------------------------------------- restricted step --
class RestrictedStep:                                    # restricted_step.py:11-124
    names = []

    def __init__(self, pes, order, delta, method='qn', tol=None, maxiter=1000,
                 d1=None, W=None):
        self.pes = pes
        self.delta = delta
        self.d1 = d1
        g0 = pes.get_g()
        self.scons = pes.get_scons()
        g = g0 + pes.get_H() @ self.scons
        stepper = method if isinstance(method, type) else get_stepper(method.lower())
        if self.cons(self.scons) - self.delta > 1e-8:    # :44-48
            self.P = pes.get_Unred().T
            self.stepper = NaiveStep(self.P @ self.scons)
            self.scons[:] *= 0
        else:
            self.P = pes.get_Ufree().T if W is None else pes.get_Ufree().T @ W   # :50-53
            if d1 is not None:                                                # :54-56
                d1 = np.linalg.lstsq(self.P.T, d1, rcond=None)[0]
            self.stepper = stepper(self.P @ g,
                                   pes.get_HL_projected(self.P.T), order, d1=d1)
        if tol is None:
            tol = 1e-10 if self.stepper.newton_safe else 1e-15
        self.tol = tol
        self.maxiter = maxiter
        self.alpha_trace = []

    def cons(self, s, dsda=None):
        raise NotImplementedError

    def eval(self, alpha):
        s, dsda = self.stepper.get_s(alpha)
        stot = self.P.T @ s + self.scons
        val, dval = self.cons(stot, self.P.T @ dsda)
        self.alpha_trace.append(alpha)
        return stot, val, dval

    def get_s(self):                                     # :78-120
        st = self.stepper
        alpha = st.alpha0
        s, val, dval = self.eval(alpha)
        if val < self.delta:
            assert val > 0.
            return s, val
        err = val - self.delta
        lower, upper = st.alphamin, st.alphamax
        for niter in range(self.maxiter):
            if abs(err) <= self.tol:
                break
            if np.nextafter(lower, upper) >= upper:
                break
            if err * st.slope > 0:
                upper = alpha
            else:
                lower = alpha
            a1 = alpha - err / dval
            if (np.isnan(a1) or a1 <= lower or a1 >= upper
                    or (niter > 4 and not st.newton_safe)):
                a2 = (lower + upper) / 2.
                if np.isinf(a2):
                    alpha = alpha + max(1, 0.5 * alpha) * np.sign(a2)
                else:
                    alpha = a2
            else:
                alpha = a1
            s, val, dval = self.eval(alpha)
            err = val - self.delta
        else:
            raise RuntimeError("Restricted step failed to converge!")
        assert val > 0
        return s, self.delta


class TrustRegionStep(RestrictedStep):                   # :127-142
    names = ['tr', 'trust region', 'trust-region', 'trust radius',
             'trust-radius']

    def cons(self, s, dsda=None):
        val = np.linalg.norm(s)
        if dsda is None:
            return val
        return val, dsda @ s / max(val, 1e-12)


class IRCTrustRegionStep(TrustRegionStep):               # :145-158
    names = []

    def __init__(self, *args, sqrtm=None, **kwargs):
        assert sqrtm is not None
        self.sqrtm = sqrtm
        self.d1 = kwargs.get('d1')
        TrustRegionStep.__init__(self, *args, **kwargs)
        assert self.d1 is not None

    def cons(self, s, dsda=None):
        s = (s + self.d1) * self.sqrtm
        if dsda is not None:
            dsda = dsda * self.sqrtm
        return TrustRegionStep.cons(self, s, dsda)


class PerAtomStep(RestrictedStep):                       # :161-183
    names = ['ras', 'restricted atomic step']

    def cons(self, s, dsda=None):
        sm = s.reshape((-1, 3))
        norms = np.linalg.norm(sm, axis=1)
        i = np.argmax(norms)
        val = norms[i]
        if dsda is None:
            return val
        return val, dsda.reshape((-1, 3))[i] @ sm[i] / max(val, 1e-12)


class MaxInternalStepOracle(RestrictedStep):             # :186-243
    """Largest weighted internal-coordinate displacement: one weight per block of coordinates, in the storage
    order translations, bonds, angles, dihedrals, other, rotations (+ cell degrees of freedom)."""
    names = ['mis', 'max internal step']

    def __init__(self, pes, *args, wx=1., wb=1., wa=1., wd=1., wo=1., wc=1., **kwargs):
        if pes.int is None:
            raise ValueError("Internal coordinates are required for the MaxInternalStep trust region method")
        self.wx, self.wb, self.wa, self.wd, self.wo, self.wc = wx, wb, wa, wd, wo, wc
        RestrictedStep.__init__(self, pes, *args, **kwargs)

    def weights(self):                                   # _get_weights :217-243
        i = self.pes.int
        w = ([self.wx] * i.ntrans + [self.wb] * i.nbonds + [self.wa] * i.nangles + [self.wd] * i.ndihedrals
             + [self.wo] * i.nother + [self.wx] * i.nrotations + [self.wc] * getattr(self.pes, 'n_cell_dof', 0))
        return np.array(w, dtype=float)

    def cons(self, s, dsda=None):                        # :206-216
        w = self.weights()
        assert len(w) == len(s)
        sw = np.abs(s * w)
        i = np.argmax(sw)
        if dsda is None:
            return sw[i]
        return sw[i], np.sign(s[i]) * dsda[i] * w[i]


def get_restricted_step(name):                           # :249-253
    for cls in (TrustRegionStep, PerAtomStep, MaxInternalStepOracle):
        if name in cls.names:
            return cls
    raise ValueError("Unknown restricted step name: {}".format(name))
