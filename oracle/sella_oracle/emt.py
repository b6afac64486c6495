"""NumPy restatement of the effective-medium-theory calculator — TEST INFRASTRUCTURE (oracle/README.md).

Functional form and parameter table of ASE's ase/calculators/emt.py (Jacobsen, Stoltze, Norskov, Surf. Sci.
366, 394 (1996)).  ASE is not part of /root/reference and not installable in the build container, so this is a
restatement of the published algorithm: **parity unpinned**; it checks the HIP kernels of
sella_amd/csrc/emt.hip and is itself validated by finite differences of its own energy.
"""
import numpy as np


class Calculator:
    def __init__(self):
        self._key, self._res, self.ncalls = None, None, 0

    def _get(self, atoms):
        key = atoms.positions.tobytes()
        if key != self._key:
            self.ncalls += 1
            self._res = self.energy_and_gradient(atoms.positions)
            self._key = key
        return self._res

    def get_potential_energy(self, atoms):
        return self._get(atoms)[0]

    def get_forces(self, atoms):
        return -self._get(atoms)[1]


class EMTOracle(Calculator):
    """Effective-medium theory in the functional form and with the parameter table of ASE's `ase.calculators.emt`
    (Jacobsen, Stoltze, Norskov, Surf. Sci. 366, 394 (1996)) for Al, Cu, Ag, Au, Ni, Pd, Pt — restated from the
    published algorithm because ASE is not installable in the build image.  **Unpinned**: it cannot be compared
    with ASE here; its forces are validated against finite differences of its own energy
    .  Periodic directions are handled by an explicit sum over the 3^d neighbouring
    images, so the cell only has to be wider than the cutoff (~5.9 A for Cu), not twice it."""
    #              E0     s0    V0     eta2   kappa  lambda n0        (eV, bohr, eV, 1/bohr, 1/bohr, 1/bohr, 1/bohr^3)
    _PAR = dict(Al=(-3.28, 3.00, 1.493, 1.240, 2.000, 1.169, 0.00700), Cu=(-3.51, 2.67, 2.476, 1.652, 2.740, 1.906, 0.00910),
                Ag=(-2.96, 3.01, 2.132, 1.652, 2.790, 1.892, 0.00547), Au=(-3.80, 3.00, 2.321, 1.674, 2.873, 2.182, 0.00703),
                Ni=(-4.44, 2.60, 3.673, 1.669, 2.757, 1.948, 0.01030), Pd=(-3.90, 2.87, 2.773, 1.818, 3.107, 2.155, 0.00688),
                Pt=(-5.85, 2.90, 4.067, 1.812, 3.145, 2.192, 0.00802))
    _BETA = 1.809                      # (16 pi / 3)^(1/3) / sqrt(2), historical rounding
    _BOHR = 0.529177210903

    def __init__(self):
        super().__init__()
        self._setup = None

    def _get(self, atoms):
        key = (tuple(atoms.symbols), np.asarray(atoms.cell, dtype=float).tobytes(), tuple(atoms.pbc))
        if self._setup is None or self._setup[0] != key:
            self._setup = (key, self._initialize(atoms))
        return super()._get(atoms)

    def _initialize(self, atoms):
        b, beta = self._BOHR, self._BETA
        kinds = sorted(set(atoms.symbols))
        par = {}
        for k in kinds:
            if k not in self._PAR:
                raise ValueError(f'EMT has no parameters for {k}')
            E0, s0, V0, eta2, kappa, lam, n0 = self._PAR[k]
            par[k] = dict(E0=E0, s0=s0 * b, V0=V0, eta2=eta2 / b, kappa=kappa / b, lam=lam / b, n0=n0 / b ** 3)
        maxseq = max(p['s0'] for p in par.values())
        rc = beta * maxseq * 0.5 * (np.sqrt(3.0) + 2.0)
        rr = rc * 2.0 * 2.0 / (np.sqrt(3.0) + 2.0)
        acut = np.log(9999.0) / (rr - rc)
        for p in par.values():
            g1 = g2 = 0.0
            for i, nn in enumerate((12, 6, 24)):
                r = p['s0'] * beta * np.sqrt(i + 1.0)
                x = nn / (12.0 * (1.0 + np.exp(acut * (r - rc))))
                g1 += x * np.exp(-p['eta2'] * (r - beta * p['s0']))
                g2 += x * np.exp(-p['kappa'] / beta * (r - beta * p['s0']))
            p['gamma1'], p['gamma2'] = g1, g2
        get = lambda name: np.array([par[s][name] for s in atoms.symbols])        # noqa: E731
        arr = {name: get(name) for name in ('E0', 's0', 'V0', 'eta2', 'kappa', 'lam', 'n0', 'gamma1', 'gamma2')}
        cell = np.asarray(atoms.cell, dtype=float)
        per = [d for d in range(3) if atoms.pbc[d]]
        shifts = [np.zeros(3)]
        for d in per:
            shifts = [sft + k * cell[d] for sft in shifts for k in (-1, 0, 1)]
        return dict(arr=arr, rc=rc, acut=acut, cutoff=rc + 0.5, shifts=np.array(shifts))

    def energy_and_gradient(self, pos):
        S = self._setup[1]
        a, beta, rc, acut = S['arr'], self._BETA, S['rc'], S['acut']
        n = len(pos)
        ii, jj, dd = [], [], []
        for sft in S['shifts']:
            d = pos[None, :, :] + sft - pos[:, None, :]              # d[i, j] = r_j + shift - r_i
            r = np.linalg.norm(d, axis=2)
            m = (r < S['cutoff']) & (r > 1e-8)
            i_, j_ = np.nonzero(m)
            ii.append(i_); jj.append(j_); dd.append(d[i_, j_])
        i, j, d = np.concatenate(ii), np.concatenate(jj), np.concatenate(dd)
        r = np.linalg.norm(d, axis=1)
        # every ordered pair (i <- j): contribution of neighbour j to atom i
        x = np.exp(acut * (r - rc))
        theta = 1.0 / (1.0 + x)
        chi = a['n0'][j] / a['n0'][i]
        dsig = np.exp(-a['eta2'][j] * (r - beta * a['s0'][j])) * chi * theta / a['gamma1'][i]
        sigma1 = np.bincount(i, weights=dsig, minlength=n)
        ypair = 0.5 * a['V0'][i] * np.exp(-a['kappa'][j] * (r / beta - a['s0'][j])) * chi / a['gamma2'][i] * theta
        e_pair = -np.bincount(i, weights=ypair, minlength=n)
        ds = -np.log(sigma1 / 12.0) / (beta * a['eta2'])
        xl = a['lam'] * ds
        yl = np.exp(-xl)
        z = 6.0 * a['V0'] * np.exp(-a['kappa'] * ds)
        e_coh = a['E0'] * ((1.0 + xl) * yl - 1.0) + z
        energy = float(np.sum(e_coh + e_pair))
        # gradient: dE/dsigma1_i * dsigma1_i/dr + d(pair)/dr, each along the pair vector
        dEdsig = -(a['E0'] * (-xl * yl) * a['lam'] + z * (-a['kappa'])) / (sigma1 * beta * a['eta2'])
        dtheta = -acut * x * theta * theta
        ddsig = dsig * (-a['eta2'][j]) + dsig / theta * dtheta
        dypair = ypair * (-a['kappa'][j] / beta) + ypair / theta * dtheta
        dEdr = dEdsig[i] * ddsig - dypair                          # derivative of the total energy w.r.t. r_ij
        u = d / r[:, None]
        g = np.zeros_like(pos)
        np.add.at(g, j, dEdr[:, None] * u)
        np.add.at(g, i, -dEdr[:, None] * u)
        return energy, g


