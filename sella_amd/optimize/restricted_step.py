"""Restricted step — the trust-radius solve behind `Sella.step` (contract of sella/optimize/restricted_step.py).

    RS(pes, order, delta, method='prfo').get_s() -> (s, smag)

picks, on a one-parameter family of steps s(alpha) (stepper.py), the member whose *measure* equals the radius
`delta` — Euclidean norm (`tr`), largest per-atom displacement (`ras`), largest weighted internal-coordinate
displacement (`mis`), mass-weighted sphere around the IRC pivot — or returns the unrestricted member if it is
already inside.

Device-first organisation (the reference evaluates a Python closure per trial alpha, each one a dense `eigh` plus
two n x m products, restricted_step.py:68-76 / stepper.py:128-157):

* the whole root find is ONE C-ABI call, `sella_restricted_step` (csrc/stepper.hip), driven through
  `BaseStepper.solve_radius`; it reproduces the reference's alpha schedule (start value, bracket, Newton steps,
  bisection from the sixth trial on for the RFO families, `nextafter` bracket test, tolerances —
  restricted_step.py:64-66, 87-117), so the trial sequence is the reference's, but a trial costs O(m) host
  arithmetic, one 2-column device matvec and one single-workgroup reduction, with two scalars coming back;
* for the trust region proper the measure is evaluated in the eigenbasis, |s + c|^2 = |shat|^2 + 2 shat.V^T c + |c|^2,
  and no device work happens per trial at all;
* the classes below only assemble the problem (constraint-corrected gradient, feasibility of the linear constraint
  step, projection basis, weights) and name the measure.  A host-side search with the same schedule remains for
  families that live on the host (`NaiveStepper`, user-supplied classes).
"""
import numpy as np

from .stepper import BaseStepper, NaiveStepper, get_stepper


def _host_search(evaluate, family, radius, tol, maxiter, trace):
    """The reference's safeguarded search (restricted_step.py:78-120) for a family evaluated on the host.
    `evaluate(alpha) -> (step, measure, d measure / d alpha)`; returns (step, reported measure)."""
    alpha = family.alpha0
    step, size, slope_a = evaluate(alpha)
    trace.append(alpha)
    if size < radius:
        return step, size
    bracket = [family.alphamin, family.alphamax]
    miss = size - radius
    for trial in range(maxiter):
        if abs(miss) <= tol or np.nextafter(bracket[0], bracket[1]) >= bracket[1]:
            return step, radius
        bracket[1 if miss * family.slope > 0 else 0] = alpha
        guess = alpha - miss / slope_a
        halve = not (bracket[0] < guess < bracket[1]) or (trial > 4 and not family.newton_safe)
        if halve:
            guess = 0.5 * (bracket[0] + bracket[1])
            if np.isinf(guess):                               # open-ended bracket of the quasi-Newton family
                guess = alpha + max(1., 0.5 * alpha) * np.sign(guess)
        alpha = guess
        step, size, slope_a = evaluate(alpha)
        trace.append(alpha)
        miss = size - radius
    raise RuntimeError("Restricted step failed to converge!")


class BaseRestrictedStep:
    synonyms = []
    measure = None                    # name of the device-side measure (device.CONSTRAINT_KINDS)

    def __init__(self, pes, order, delta, method='qn', tol=None, maxiter=1000, d1=None, W=None):
        self.pes, self.delta, self.d1, self.maxiter = pes, delta, d1, maxiter
        self.alphas = []
        family = method if isinstance(method, type) and issubclass(method, BaseStepper) else get_stepper(method.lower())
        # gradient as seen after the linear constraint correction (restricted_step.py:33-37)
        self.scons = pes.get_scons()
        # (no matrix pass for the usual case of satisfied constraints: H @ 0)
        g = pes.get_g() + (pes.get_H() @ self.scons if np.any(self.scons) else 0.0)
        self._lift = None
        self._orthonormal = W is None
        if self.cons(self.scons) - delta > 1e-8:
            # the correction alone leaves the region: move along it only (restricted_step.py:44-48)
            self._lift = pes.get_Unred()
            self.stepper = NaiveStepper(self._lift.T @ self.scons)
            self.scons = np.zeros_like(self.scons)
        else:
            basis = self._weighted(pes.get_Ufree(), W)
            extra = {} if d1 is None else dict(d1=np.linalg.lstsq(basis, d1, rcond=None)[0])      # :54-56
            # the family composes `basis` with the eigenbasis on the device and hands back unprojected vectors
            self.stepper = family(g, pes.get_HL_projected(basis), order, U=basis, **extra)
        self.tol = tol if tol is not None else (1e-10 if self.stepper.newton_safe else 1e-15)

    @staticmethod
    def _weighted(Ufree, W):
        """P^T = W^T Ufree of restricted_step.py:50-53; a diagonal W (the IRC's mass weighting) is a row scaling."""
        if W is None:
            return Ufree
        W = np.asarray(W, dtype=np.float64)
        if W.ndim == 1:
            return W[:, None] * Ufree
        if not np.any(W - np.diag(W.diagonal())):
            return W.diagonal()[:, None] * Ufree
        return W.T @ Ufree

    # ---- measure of a total step (host form; the device form lives in rs_cons_kernel) ---------------------------
    def cons(self, s, dsda=None):
        raise NotImplementedError

    def _measure_args(self):
        return {}

    def eval(self, alpha):
        """One trial (host evaluation of the measure; kept for callers that probe the family themselves)."""
        s, dsda = self.stepper.get_s(alpha)
        if self._lift is not None:
            s, dsda = self._lift @ s, self._lift @ dsda
        total = s + self.scons
        size, dsize = self.cons(total, dsda)
        self.alphas.append(alpha)
        return total, size, dsize

    def _device_search_applies(self):
        """The one-call device search evaluates the BUILT-IN families and measures.  A user-supplied stepper that
        overrides `get_s` (or never ran `BaseStepper._stepper_init`), a restricted-step subclass with its own `cons`,
        or one without a named measure is searched on the host with the same schedule."""
        st = self.stepper
        if self._lift is not None or self.measure is None or getattr(st, '_dev', None) is None:
            return False
        if getattr(type(st), 'get_s', None) is not BaseStepper.get_s:
            return False
        return any(type(self).cons is base.cons for base in _builtin_measures())

    def get_s(self):
        if self._device_search_applies():
            s, size, trials = self.stepper.solve_radius(self.measure, self.delta, self.tol, self.maxiter,
                                                        scons=self.scons, orthonormal=self._orthonormal,
                                                        **self._measure_args())
            self.alphas = list(trials)
            return s, size
        del self.alphas[:]
        trace = []
        out = _host_search(lambda a: self.eval(a), self.stepper, self.delta, self.tol, self.maxiter, trace)
        self.alphas = trace
        return out

    @classmethod
    def match(cls, name):
        return name in cls.synonyms


class TrustRegion(BaseRestrictedStep):
    synonyms = ['tr', 'trust region', 'trust-region', 'trust radius', 'trust-radius']
    measure = 'tr'

    def cons(self, s, dsda=None):
        size = float(np.sqrt(s @ s))
        return size if dsda is None else (size, float(dsda @ s) / max(size, 1e-12))


class IRCTrustRegion(TrustRegion):
    """Sphere of radius dx in mass-weighted coordinates around the pivot of the current IRC step:
    |(s + d1) * sqrt(m)| (restricted_step.py:145-158)."""
    synonyms = []
    measure = 'sphere'

    def __init__(self, *args, sqrtm=None, **kwargs):
        if sqrtm is None or kwargs.get('d1') is None:
            raise ValueError('IRCTrustRegion needs sqrtm and the accumulated displacement d1')
        self.sqrtm = np.asarray(sqrtm, dtype=np.float64)
        self.d1 = kwargs['d1']
        TrustRegion.__init__(self, *args, **kwargs)

    def cons(self, s, dsda=None):
        return TrustRegion.cons(self, (s + self.d1) * self.sqrtm, None if dsda is None else dsda * self.sqrtm)

    def _measure_args(self):
        return dict(w=self.sqrtm, d1=self.d1)


class RestrictedAtomicStep(BaseRestrictedStep):
    synonyms = ['ras', 'restricted atomic step']
    measure = 'ras'

    def __init__(self, pes, *args, **kwargs):
        if pes.int is not None:
            raise ValueError("Internal coordinates are not compatible with "
                             f"the {self.__class__.__name__} trust region method.")
        BaseRestrictedStep.__init__(self, pes, *args, **kwargs)

    def cons(self, s, dsda=None):
        per_atom = s.reshape((-1, 3))
        sizes = np.sqrt(np.einsum('ij,ij->i', per_atom, per_atom))
        worst = int(sizes.argmax())
        if dsda is None:
            return sizes[worst]
        return sizes[worst], float(dsda.reshape((-1, 3))[worst] @ per_atom[worst]) / max(sizes[worst], 1e-12)


class MaxInternalStep(BaseRestrictedStep):
    synonyms = ['mis', 'max internal step']
    measure = 'mis'
    _kinds = (('wx', 'ntrans'), ('wb', 'nbonds'), ('wa', 'nangles'), ('wd', 'ndihedrals'), ('wo', 'nother'),
              ('wx', 'nrotations'))                      # weight attribute per block of coordinates, in storage order

    def __init__(self, pes, *args, wx=1., wb=1., wa=1., wd=1., wo=1., wc=1., **kwargs):
        if pes.int is None:
            raise ValueError("Internal coordinates are required for the "
                             f"{self.__class__.__name__} trust region method")
        self.wx, self.wb, self.wa, self.wd, self.wo, self.wc = wx, wb, wa, wd, wo, wc
        BaseRestrictedStep.__init__(self, pes, *args, **kwargs)

    def _get_weights(self):
        counts = [(getattr(self, wname), getattr(self.pes.int, cname)) for wname, cname in self._kinds]
        counts.append((self.wc, getattr(self.pes, 'n_cell_dof', 0)))
        return np.repeat([wt for wt, _ in counts], [cnt for _, cnt in counts]).astype(np.float64)

    def cons(self, s, dsda=None):
        w = self._get_weights()
        if len(w) != len(s):
            raise ValueError(f'{len(w)} coordinate weights for a step of length {len(s)}')
        scaled = np.abs(s * w)
        worst = int(scaled.argmax())
        if dsda is None:
            return scaled[worst]
        return scaled[worst], np.sign(s[worst]) * dsda[worst] * w[worst]

    def _measure_args(self):
        return dict(w=self._get_weights())


_all_restricted_step = [TrustRegion, RestrictedAtomicStep, MaxInternalStep]


def _builtin_measures():
    return (TrustRegion, IRCTrustRegion, RestrictedAtomicStep, MaxInternalStep)


def get_restricted_step(name):
    for candidate in _all_restricted_step:
        if candidate.match(name):
            return candidate
    raise ValueError("Unknown restricted step name: {}".format(name))
