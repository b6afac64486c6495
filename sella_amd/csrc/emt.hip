// emt.hip — effective-medium-theory energy and forces (the calculator on the far side of PES.eval,
// sella/peswrapper.py:413-418).  Functional form and parameter handling of
// ASE's ase/calculators/emt.py (Jacobsen, Stoltze, Norskov, Surf. Sci. 366, 394 (1996)); ASE is not part of
// /root/reference, so this is a restatement of the published algorithm, checked against the NumPy
// restatement in oracle/ and against finite differences of its own energy.
//
// All-pairs formulation: one workgroup per atom sweeps every (neighbour, periodic image) pair inside the
// cutoff — 1024 atoms x 9 images are 9.4 M pair terms per pass, far below anything a neighbour list would
// pay for itself on this device — and reduces in-block, so energies and forces are deterministic (no atomics).
//   pass 1: sigma1_i = sum_j dsigma(i <- j), pair energy of atom i
//           cohesive function, dE/dsigma1_i                           (own density only: at the end of pass 1)
//   pass 2: F_i by gathering both ordered pairs (i <- j) and (j <- i) of every neighbour
#include "internal.h"

namespace sella {
namespace {

struct EmtPar {             // per-atom parameters, already converted to eV / Angstrom
    const double *E0, *s0, *V0, *eta2, *kappa, *lam, *n0, *gamma1, *gamma2;
};

struct EmtArgs {
    int n, nshift;
    int hcap;               // slots of a thread's neighbour list in use (<= EMT_HCAP; option emt_hcap, tests lower it)
    const double* pos;      // n x 3
    const double* shifts;   // nshift x 3 lattice translations (including 0)
    EmtPar p;
    double rc, acut, cutoff, beta;
    double* sigma1; double* epair; double* dEdsig; double* eatom; double* grad;
    int* nbr;               // n x 256 x (1 + EMT_HCAP): neighbour lists of the density kernel's threads, for the force kernel
};

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// The sweep over all (neighbour, image) pairs is split in two per thread: first the distance test alone over the thread's
// pairs (t = tid, tid + 256, ...: no division, squared distance against a slightly widened cutoff), the few pairs inside
// the cutoff noted in LDS; then the exponentials for those only.  Under 1 % of the pairs are neighbours, but with the
// expensive branch inside the sweep every second wavefront iteration took it for some lane; deferred, a wavefront
// pays for the largest number of neighbours any of its lanes found (2-3).  The terms are added per thread in the same
// order as before — sums are bit-identical to the one-loop form.
constexpr int EMT_HCAP = 8;
constexpr int EMT_LDS_ATOMS = 1024;       // up to this many atoms the positions are staged in LDS (structure of arrays)

// a noted pair: atom index in the low 24 bits, image above (no division when it is taken up again)
__device__ __forceinline__ int emt_pack(int j, int s) { return (s << 24) | j; }

struct EmtStage {
    double x[EMT_LDS_ATOMS], y[EMT_LDS_ATOMS], z[EMT_LDS_ATOMS];
};

// All pairs of this thread (t = s n + j = tid mod 256, in increasing t) through `heavy`, in two steps: the distance test alone
// — image by image (s uniform: its shift sits in scalar registers), neighbours noted in hits[tid][..] — then the noted pairs.
// A thread whose list fills up (never at EMT's cutoff and 256 threads) works it off and takes its remaining pairs
// directly, in the same order; the count returned is then negative: the stored list is incomplete.
template <bool STAGED, class Heavy>
__device__ __forceinline__ int emt_pairs_impl(const EmtArgs& a, double xi, double yi, double zi, const EmtStage* st,
                                              int (*hits)[EMT_HCAP + 1], Heavy heavy) {
    const int tid = threadIdx.x, n = a.n;
    const double cut2 = a.cutoff * a.cutoff * (1.0 + 1e-12);
    const bool aligned = (n & 255) == 0;
    auto first_j = [&](int s) { return aligned ? tid : (((tid - s * n) % 256) + 256) % 256; };
    // (pos + shift) - x_i as in the terms themselves: the same rounding decides which pairs are neighbours
    auto near = [&](int j, double shx, double shy, double shz) {
        const double px = STAGED ? st->x[j] : a.pos[3 * j], py = STAGED ? st->y[j] : a.pos[3 * j + 1];
        const double pz = STAGED ? st->z[j] : a.pos[3 * j + 2];
        const double dx = px + shx - xi, dy = py + shy - yi, dz = pz + shz - zi;
        return dx * dx + dy * dy + dz * dz <= cut2;
    };
    int nh = 0, rs = -1, rj = 0;
    for (int s = 0; s < a.nshift; ++s) {
        const double shx = a.shifts[3 * s], shy = a.shifts[3 * s + 1], shz = a.shifts[3 * s + 2];
        if (rs >= 0) continue;                                     // (this thread's list is full: see below)
        for (int j = first_j(s); j < n; j += 1024) {
            bool in[4];                                            // four tests at a time: their loads are in flight together
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ju = j + 256 * u;
                in[u] = near(ju < n ? ju : j, shx, shy, shz) && ju < n;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (in[u] && rs < 0) {
                    if (nh == a.hcap) { rs = s; rj = j + 256 * u; }
                    else hits[tid][nh++] = emt_pack(j + 256 * u, s);
                }
            }
            if (rs >= 0) break;
        }
    }
    for (int h = 0; h < nh; ++h) heavy(hits[tid][h]);
    if (rs < 0) return nh;
    for (int s = rs; s < a.nshift; ++s) {
        const double shx = a.shifts[3 * s], shy = a.shifts[3 * s + 1], shz = a.shifts[3 * s + 2];
        for (int j = (s == rs) ? rj : first_j(s); j < n; j += 256)
            if (near(j, shx, shy, shz)) heavy(emt_pack(j, s));
    }
    return -1;
}

template <class Heavy>
__device__ __forceinline__ int emt_pairs(const EmtArgs& a, double xi, double yi, double zi, EmtStage* st, int (*hits)[EMT_HCAP + 1],
                                         Heavy heavy) {
    if (a.n <= EMT_LDS_ATOMS) {
        for (int j = threadIdx.x; j < a.n; j += 256) { st->x[j] = a.pos[3 * j]; st->y[j] = a.pos[3 * j + 1]; st->z[j] = a.pos[3 * j + 2]; }
        __syncthreads();
        return emt_pairs_impl<true>(a, xi, yi, zi, st, hits, heavy);
    }
    return emt_pairs_impl<false>(a, xi, yi, zi, st, hits, heavy);
}

__device__ __forceinline__ void emt_density_vb(const VB vb, EmtArgs a) {
    __shared__ double red[4];
    __shared__ int hits[256][EMT_HCAP + 1];
    __shared__ EmtStage stage;
    const int i = vb.x;
    const double xi = a.pos[3 * i], yi = a.pos[3 * i + 1], zi = a.pos[3 * i + 2];
    const double n0i = a.p.n0[i], g1i = a.p.gamma1[i], g2i = a.p.gamma2[i], V0i = a.p.V0[i];
    double sig = 0.0, ep = 0.0;
    auto heavy = [&](int t) {
        const int j = t & 0xffffff, s = t >> 24;
        const double dx = a.pos[3 * j] + a.shifts[3 * s] - xi;
        const double dy = a.pos[3 * j + 1] + a.shifts[3 * s + 1] - yi;
        const double dz = a.pos[3 * j + 2] + a.shifts[3 * s + 2] - zi;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        if (r < a.cutoff && r > 1e-8) {
            const double theta = 1.0 / (1.0 + exp(a.acut * (r - a.rc)));
            const double chi = a.p.n0[j] / n0i;
            sig += exp(-a.p.eta2[j] * (r - a.beta * a.p.s0[j])) * chi * theta / g1i;
            ep += 0.5 * V0i * exp(-a.p.kappa[j] * (r / a.beta - a.p.s0[j])) * chi / g2i * theta;
        }
    };
    const int nh = emt_pairs(a, xi, yi, zi, &stage, hits, heavy);
    // the neighbour lists go on to the force kernel (same atom, same thread, same order): count < 0 = incomplete
    {
        int* out = a.nbr + ((size_t)i * 256 + threadIdx.x) * (EMT_HCAP + 1);
        out[0] = nh;
        for (int h = 0; h < nh; ++h) out[1 + h] = hits[threadIdx.x][h];
    }
    sig = block_sum(sig, red);
    ep = block_sum(ep, red);
    if (threadIdx.x == 0) {
        a.sigma1[i] = sig;
        a.epair[i] = -ep;
        // cohesive function and dE/dsigma1 of this atom (own density only: no pass of its own)
        const double ds = -log(sig / 12.0) / (a.beta * a.p.eta2[i]);
        const double xl = a.p.lam[i] * ds, yl = exp(-xl);
        const double z = 6.0 * a.p.V0[i] * exp(-a.p.kappa[i] * ds);
        a.eatom[i] = a.p.E0[i] * ((1.0 + xl) * yl - 1.0) + z + -ep;
        a.dEdsig[i] = (a.p.E0[i] * xl * yl * a.p.lam[i] + z * a.p.kappa[i]) / (sig * a.beta * a.p.eta2[i]);
    }
}
__global__ __launch_bounds__(256) void emt_density_kernel(EmtArgs a) { emt_density_vb(vb_hw(), a); }

__device__ __forceinline__ void emt_force_vb(const VB vb, EmtArgs a) {
    __shared__ double red[4];
    __shared__ int hits[256][EMT_HCAP + 1];
    __shared__ EmtStage stage;
    __shared__ int incomplete;
    const int i = vb.x;
    const double xi = a.pos[3 * i], yi = a.pos[3 * i + 1], zi = a.pos[3 * i + 2];
    const double n0i = a.p.n0[i], g1i = a.p.gamma1[i], g2i = a.p.gamma2[i], V0i = a.p.V0[i];
    const double eta2i = a.p.eta2[i], kapi = a.p.kappa[i], s0i = a.p.s0[i], dEi = a.dEdsig[i];
    double gx = 0.0, gy = 0.0, gz = 0.0;
    auto heavy = [&](int t) {
        const int j = t & 0xffffff, s = t >> 24;
        const double dx = a.pos[3 * j] + a.shifts[3 * s] - xi;
        const double dy = a.pos[3 * j + 1] + a.shifts[3 * s + 1] - yi;
        const double dz = a.pos[3 * j + 2] + a.shifts[3 * s + 2] - zi;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        if (r < a.cutoff && r > 1e-8) {
            const double x = exp(a.acut * (r - a.rc));
            const double theta = 1.0 / (1.0 + x);
            const double dth = -a.acut * x * theta;             // d(theta)/dr / theta
            const double chi = a.p.n0[j] / n0i;
            // ordered pair (i <- j): neighbour j seen from i
            const double dsig_ij = exp(-a.p.eta2[j] * (r - a.beta * a.p.s0[j])) * chi * theta / g1i;
            const double y_ij = 0.5 * V0i * exp(-a.p.kappa[j] * (r / a.beta - a.p.s0[j])) * chi / g2i * theta;
            // ordered pair (j <- i): this atom seen from j (same distance, opposite direction)
            const double dsig_ji = exp(-eta2i * (r - a.beta * s0i)) / chi * theta / a.p.gamma1[j];
            const double y_ji = 0.5 * a.p.V0[j] * exp(-kapi * (r / a.beta - s0i)) / chi / a.p.gamma2[j] * theta;
            // dE/dr of this pair distance, both ordered pairs:  dE/dsigma1 * dsigma/dr - d(pair energy)/dr
            const double dEdr = dEi * dsig_ij * (-a.p.eta2[j] + dth) - y_ij * (-a.p.kappa[j] / a.beta + dth)
                                + a.dEdsig[j] * dsig_ji * (-eta2i + dth) - y_ji * (-kapi / a.beta + dth);
            const double f = dEdr / r;                          // dr/dx_i = -(d / r)
            gx -= f * dx;
            gy -= f * dy;
            gz -= f * dz;
        }
    };
    // neighbours from the density kernel's lists; the sweep again only if some thread's list overflowed there
    const int* lst = a.nbr + ((size_t)i * 256 + threadIdx.x) * (EMT_HCAP + 1);
    const int cnt = lst[0];
    if (threadIdx.x == 0) incomplete = 0;
    __syncthreads();
    if (cnt < 0) incomplete = 1;
    __syncthreads();
    if (incomplete) {
        (void)emt_pairs(a, xi, yi, zi, &stage, hits, heavy);
    } else {
        for (int h = 0; h < cnt; ++h) heavy(lst[1 + h]);
    }
    gx = block_sum(gx, red);
    gy = block_sum(gy, red);
    gz = block_sum(gz, red);
    if (threadIdx.x == 0) {
        a.grad[3 * i] = gx;
        a.grad[3 * i + 1] = gy;
        a.grad[3 * i + 2] = gz;
    }
}
__global__ __launch_bounds__(256) void emt_force_kernel(EmtArgs a) { emt_force_vb(vb_hw(), a); }

}  // namespace
}  // namespace sella

using namespace sella;

// dconst != nullptr: the parameter table and the shift vectors are resident already (9 n + 3 nshift doubles, uploaded by
// the caller once: sella_calc_emt_create) — a force call is then one upload (positions), two kernels, one read-back
int sella::emt_eval_resident(sella_ctx* c, int n, const double* pos, const double* par, int nshift, const double* shifts,
                             const double* dconst, double rc, double acut, double cutoff, double beta, double* energy,
                             double* grad) {
    double *dea, *dgr;
    SCHK(emt_queue(c, n, pos, par, nshift, shifts, dconst, rc, acut, cutoff, beta, &dea, &dgr));
    // per-atom energies and the gradient sit back to back: one read-back
    std::vector<double>& out = c->hbuf_b;
    out.resize((size_t)4 * n);
    SCHK(d2h_async(c, out.data(), dea, (size_t)4 * n * sizeof(double)));
    SCHK(stream_wait(c));
    double e = 0.0;
    for (int i = 0; i < n; ++i) e += out[i];
    *energy = e;
    for (size_t i = 0; i < (size_t)3 * n; ++i) grad[i] = out[(size_t)n + i];
    return SELLA_OK;
}

// the same up to the kernels: positions uploaded, two launches queued, nothing waited for.  *eatom (n per-atom
// energies, summed by the caller in index order) and *grad (3 n, directly behind) stay valid until scratch slot
// SCR_MISC0 is used again.
int sella::emt_queue(sella_ctx* c, int n, const double* pos, const double* par, int nshift, const double* shifts,
                     const double* dconst, double rc, double acut, double cutoff, double beta, double** eatom, double** grad) {
    if (n >= (1 << 24) || nshift > 127) { set_error("emt: at most 2^24 atoms and 127 periodic images"); return SELLA_E_INVALID; }
    const size_t nbr_words = ((size_t)n * 256 * (EMT_HCAP + 1) + 1) / 2;
    const size_t words = (size_t)3 * n + (size_t)9 * n + (size_t)3 * nshift + (size_t)4 * n + (size_t)3 * n + 64 + nbr_words;
    double* buf;
    SCHK(scratch_get(c, SCR_MISC0, words * sizeof(double), &buf));
    double* dpos = buf;
    double* dpar = dpos + 3 * (size_t)n;
    double* dsh = dpar + 9 * (size_t)n;
    double* dsig = dsh + 3 * (size_t)nshift;
    double* dep = dsig + n;
    double* dde = dep + n;
    double* dea = dde + n;
    double* dgr = dea + n;
    SCHK(h2d_async(c, dpos, pos, (size_t)3 * n * sizeof(double)));
    if (dconst) {
        dpar = const_cast<double*>(dconst);
        dsh = dpar + 9 * (size_t)n;
    } else {
        SCHK(h2d_async(c, dpar, par, (size_t)9 * n * sizeof(double)));
        SCHK(h2d_async(c, dsh, shifts, (size_t)3 * nshift * sizeof(double)));
    }
    EmtArgs a;
    a.n = n; a.nshift = nshift; a.pos = dpos; a.shifts = dsh;
    a.hcap = (int)std::min<long>(std::max<long>(c->opt.emt_hcap, 1), EMT_HCAP);
    a.p.E0 = dpar; a.p.s0 = dpar + n; a.p.V0 = dpar + 2 * (size_t)n; a.p.eta2 = dpar + 3 * (size_t)n;
    a.p.kappa = dpar + 4 * (size_t)n; a.p.lam = dpar + 5 * (size_t)n; a.p.n0 = dpar + 6 * (size_t)n;
    a.p.gamma1 = dpar + 7 * (size_t)n; a.p.gamma2 = dpar + 8 * (size_t)n;
    a.rc = rc; a.acut = acut; a.cutoff = cutoff; a.beta = beta;
    a.sigma1 = dsig; a.epair = dep; a.dEdsig = dde; a.eatom = dea; a.grad = dgr;
    a.nbr = reinterpret_cast<int*>(dgr + 3 * (size_t)n + 32);
    SELLA_LAUNCHB(c, emt_density_kernel, emt_density_vb, 256, dim3(n), dim3(256), 0, a);
    SELLA_LAUNCHB(c, emt_force_kernel, emt_force_vb, 256, dim3(n), dim3(256), 0, a);
    HIPCHK(hipGetLastError());
    *eatom = dea;
    *grad = dgr;
    return SELLA_OK;
}

extern "C" int sella_emt_eval(sella_ctx* c, int n, const double* pos, const double* par /* 9 x n */, int nshift,
                              const double* shifts, double rc, double acut, double cutoff, double beta,
                              double* energy, double* grad) {
    if (!c || n <= 0 || !pos || !par || nshift <= 0 || !shifts || !energy || !grad) {
        set_error("emt_eval: invalid arguments");
        return SELLA_E_INVALID;
    }
    return emt_eval_resident(c, n, pos, par, nshift, shifts, nullptr, rc, acut, cutoff, beta, energy, grad);
}
