// A whole saddle-point search as library calls: `Sella.run` (sella/optimize/optimize.py:317-440 driven by ASE's
// Optimizer.irun) for the configuration the ensemble of BASELINE configs[3] consists of — Cartesian PES
// (sella/peswrapper.py:214-607), no constraints or constraints that pin single coordinates, a calculator that lives in the
// library (calc.hip), approximate Hessian in structured form (eigh.hip / lrstep.hip), built-in step family and measure.
//
// Nothing here is new arithmetic: the loop strings together the entry points the host-language driver calls —
// sella_davidson over sella_fd_matvec (PES.diag, peswrapper.py:508-556), sella_symmetrize_y and the secant-pair
// rotation of :545-553, sella_update_h_lr for the block update (linalg.py:274-304), sella_lr_restrict for the view of
// pinned coordinates (peswrapper.py:363-386), sella_opt_step for every optimizer step — with the reference's schedule
// (first-use diagonalisation, optimize.py:318-326; re-diagonalisation rule, :363-378; convergence test of
// peswrapper.py:438-441) in between.  What it removes is the interpreter: a 3N = 768 member spends ~10,000 host-language
// function calls per search, a third of its wall time, and — holding the interpreter lock — keeps host threads from
// sharing a GPU.  With the search in the library one thread per replica scales like the launch-and-wait loops of
// tools/lab/wait_lab.hip.
#include "internal.h"

#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

using namespace sella;

struct sella_search {
    sella_ctx* c = nullptr;
    sella_calc* calc = nullptr;
    int n = 0, m = 0;
    std::vector<int> idx;                       // free coordinates (empty: all n)
    sella_search_params_t p;
    // point
    std::vector<double> x, g, gold, s, dx, target;
    double f = 0.0, delta = 0.0, rho = 1.0, smag = 0.0;
    bool have_fg = false, initialized = false, first_diag = true, have_step = false, released = false;
    long nsteps_since_diag = 0, nsteps = 0, neval = 0, nfused = 0;
    // approximate Hessian: structured form (+ view of the free coordinates)
    bool H_none = true, have_view = false;
    sella_mat B = SELLA_NO_MAT, Wt = SELLA_NO_MAT, Bsub = SELLA_NO_MAT, Wt_sub = SELLA_NO_MAT;
    int r = 0, r_sub = 0, cap = 0, cap_sub = 0, rank_limit = 0, rank_limit_sub = 0;
    std::vector<double> mu, mu_sub;
    double lam0 = 0.0;
    int B_stale = 0, Bsub_stale = 0;
    sella_opt_step_t io = sella_opt_step_t();   // (persistent: its alpha_hint carries from step to step)
    // Hand-over (SELLA_E_UNSUPPORTED) in the middle of a step: a diagonalisation whose block of secant pairs no longer
    // fits the structured form has spent its force calls — the pairs are kept for the caller, who applies them on the
    // dense route (sella_search_pending_pairs), and the step that scheduled it counts.
    std::vector<double> pendS, pendY;           // (n x pend_k) row-major each
    int pend_k = 0;
    bool count_step_on_exit = false;
};

namespace {

int rank_limit_of(int dim) { return std::max(std::max(1, (int)(0.4 * dim)), 8); }      // linalg.py LR_MAX_FRACTION

int evaluate(sella_search* S) {
    SCHK(sella_calc_eval(S->calc, S->x.data(), &S->f, S->g.data()));
    ++S->neval;
    S->have_fg = true;
    return SELLA_OK;
}

// peswrapper.py:438-441: largest per-atom norm of the projected forces (pinned coordinates carry none)
double fmax_now(const sella_search* S) {
    static thread_local std::vector<char> freec;           // (one search per thread at a time: rebuilt per call only if pinned)
    freec.clear();
    if (!S->idx.empty()) {
        freec.assign(S->n, 0);
        for (int q : S->idx) freec[q] = 1;
    }
    double best = 0.0;
    for (int a = 0; a + 2 < S->n; a += 3) {
        double s2 = 0.0;
        for (int d = 0; d < 3; ++d) {
            const double v = (freec.empty() || freec[a + d]) ? S->g[a + d] : 0.0;
            s2 += v * v;
        }
        best = std::max(best, std::sqrt(s2));
    }
    return best;
}

int ensure_view(sella_search* S) {
    if (S->idx.empty() || S->have_view || S->H_none) return SELLA_OK;
    if (S->r > 256) { set_error("search: explicit rank %d too large for a structured view", S->r); return SELLA_E_UNSUPPORTED; }
    SCHK(sella_mat_alloc(S->c, S->cap_sub, S->m, &S->Wt_sub));
    SCHK(sella_mat_alloc(S->c, S->m, S->m, &S->Bsub));
    S->mu_sub.assign((size_t)S->cap_sub, 0.0);
    SCHK(sella_lr_restrict(S->c, S->Wt, S->r, S->mu.data(), S->lam0, S->idx.data(), S->m, S->Wt_sub, &S->r_sub,
                           S->mu_sub.data()));
    S->Bsub_stale = 1;                           // the matrix itself is rebuilt from the decomposition if ever needed
    S->have_view = true;
    return SELLA_OK;
}

// ApproximateHessian.update with a block of k secant pairs (linalg.py:274-304)
int update_block(sella_search* S, const double* Sm, const double* Ym, int k) {
    sella_ctx* c = S->c;
    const int n = S->n;
    if (S->H_none) {
        if (2 * k > rank_limit_of(n)) { set_error("search: first update of rank %d exceeds the structured form", 2 * k); return SELLA_E_UNSUPPORTED; }
        // B = lam0 I + update, lam0 the geometric mean |Ritz value| of S^T Ytilde (hessian_update.py:58-67)
        std::vector<double> Yt((size_t)n * k), M((size_t)k * k, 0.0), th(k), Z((size_t)k * k), work(k);
        if (k == 1 || S->p.symm < 0) Yt.assign(Ym, Ym + (size_t)n * k);
        else SCHK(sella_symmetrize_y(c, Sm, Ym, n, k, S->p.symm, Yt.data()));
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < k; ++a) {
                const double sa = Sm[(size_t)i * k + a];
                for (int b = 0; b < k; ++b) M[(size_t)a * k + b] += sa * Yt[(size_t)i * k + b];
            }
        if (small::sym_eig(k, M.data(), k, th.data(), Z.data(), k, work.data()) != 0) {
            set_error("search: small eigenproblem did not converge");
            return SELLA_E_NOCONV;
        }
        double acc = 0.0;
        for (int a = 0; a < k; ++a) acc += std::log(std::max(std::fabs(th[a]), 1e-12));
        S->lam0 = std::exp(acc / k);
        SCHK(sella_mat_alloc(c, n, n, &S->B));
        SCHK(sella_mat_add_diag(c, S->B, S->lam0));
        SCHK(sella_mat_alloc(c, S->cap, n, &S->Wt));
        S->mu.assign((size_t)S->cap, 0.0);
        S->r = 0;
        int nr1 = 0;
        SCHK(sella_update_h_lr(c, S->B, S->Wt, &S->r, S->mu.data(), S->lam0, Sm, Ym, n, k, S->p.update_method, S->p.symm, &nr1,
                               SELLA_NO_MAT, SELLA_NO_MAT, nullptr, nullptr, nullptr, 0, nullptr));
        S->H_none = false;
        S->B_stale = 0;
        return SELLA_OK;
    }
    if (S->r + 2 * k > S->rank_limit || S->r + 2 * k + 4 > S->cap) {
        set_error("search: explicit rank %d + %d leaves the structured form", S->r, 2 * k);
        return SELLA_E_UNSUPPORTED;
    }
    if (S->B_stale) { SCHK(sella_lr_materialize(c, S->B, S->Wt, S->r, S->mu.data(), S->lam0)); S->B_stale = 0; }
    int nr1 = 0, nr1s = 0;
    if (S->have_view) {
        if (S->r_sub + 2 * k > S->rank_limit_sub || S->r_sub + 2 * k + 4 > S->cap_sub) {
            set_error("search: explicit rank of the view %d + %d leaves the structured form", S->r_sub, 2 * k);
            return SELLA_E_UNSUPPORTED;
        }
        if (S->Bsub_stale) {
            SCHK(sella_lr_materialize(c, S->Bsub, S->Wt_sub, S->r_sub, S->mu_sub.data(), S->lam0));
            S->Bsub_stale = 0;
        }
        return sella_update_h_lr(c, S->B, S->Wt, &S->r, S->mu.data(), S->lam0, Sm, Ym, n, k, S->p.update_method, S->p.symm, &nr1,
                                 S->Bsub, S->Wt_sub, &S->r_sub, S->mu_sub.data(), S->idx.data(), S->m, &nr1s);
    }
    return sella_update_h_lr(c, S->B, S->Wt, &S->r, S->mu.data(), S->lam0, Sm, Ym, n, k, S->p.update_method, S->p.symm, &nr1,
                             SELLA_NO_MAT, SELLA_NO_MAT, nullptr, nullptr, nullptr, 0, nullptr);
}

// PES.diag (peswrapper.py:508-556): Davidson on the finite-difference Hessian in the free coordinates, preconditioned
// by the projected approximate Hessian; every product becomes a secant pair of the block update afterwards
int diagonalise(sella_search* S) {
    sella_ctx* c = S->c;
    const int n = S->n, m = S->m;
    if (m == 0) return SELLA_OK;
    SCHK(ensure_view(S));
    const bool pins = !S->idx.empty();
    // preconditioner: structured decomposition of the projected Hessian
    sella_mat hP = SELLA_NO_MAT, hPt = SELLA_NO_MAT;
    const double* pevals = nullptr;
    double pscale = 1.0;
    int rP = 0;
    if (!S->H_none) {
        const sella_mat src = pins ? S->Wt_sub : S->Wt;
        rP = pins ? S->r_sub : S->r;
        pevals = pins ? S->mu_sub.data() : S->mu.data();
        pscale = S->lam0;
        if (rP > 0) {
            SCHK(sella_mat_rows(c, src, 0, rP, &hPt));
            int st = sella_mat_transpose(c, hPt, &hP);
            if (st != SELLA_OK) { sella_mat_free(c, hPt); return st; }
        }
    }
    auto cleanup = [&](int code) {
        if (hP != SELLA_NO_MAT) sella_mat_free(c, hP);
        if (hPt != SELLA_NO_MAT) sella_mat_free(c, hPt);
        return code;
    };
    // start block (peswrapper.py:518-525, eigensolvers.py:43-50)
    std::vector<double> start;
    int nv0 = 0;
    if (S->H_none || S->first_diag) {
        double nrm = 0.0;
        start.resize((size_t)m);
        for (int q = 0; q < m; ++q) { start[q] = S->g[pins ? S->idx[q] : q]; nrm += start[q] * start[q]; }
        nv0 = std::sqrt(nrm) < 1e-12 ? 0 : 1;
    }
    if (nv0 == 0) {
        if (rP > 0) {
            int nneg = 0;
            for (int i = 0; i < rP; ++i) nneg += pevals[i] < 0.0 ? 1 : 0;
            nneg = std::max(1, nneg);
            std::vector<double> rows((size_t)rP * m);
            int st = sella_mat_download(c, hPt, rows.data());
            if (st != SELLA_OK) return cleanup(st);
            start.assign((size_t)m * nneg, 0.0);
            for (int j = 0; j < nneg; ++j)
                for (int i = 0; i < m; ++i) start[(size_t)i * nneg + j] = rows[(size_t)j * m + i];
            nv0 = nneg;
        } else {
            start.assign((size_t)m, 0.0);
            start[0] = 1.0;
            nv0 = 1;
        }
    }
    sella_fd* fd = nullptr;
    {
        int st = sella_fd_create(S->calc, n, S->x.data(), S->g.data(), S->p.eta, S->p.threepoint, pins ? S->idx.data() : nullptr,
                                 pins ? m : 0, &fd);
        if (st != SELLA_OK) return cleanup(st);
    }
    const int maxiter = 2 * m + 1;
    const int kmax = std::min(m, std::max(maxiter, nv0));
    // (capacity for the largest subspace the solver may return, m x (kmax + 1) doubles twice — 2.4 MB for 384 free
    //  coordinates, of which a run uses some twenty columns: NOT value-initialised, so only the pages the solver writes are
    //  ever touched; a zero-filled std::vector cost 0.34 ms of page faults per diagonalisation)
    std::vector<double> lams((size_t)kmax + 1);
    std::unique_ptr<double[]> V(new double[(size_t)m * (kmax + 1)]), AV(new double[(size_t)m * (kmax + 1)]);
    int k = 0, nmv = 0;
    int st = sella_davidson(c, SELLA_NO_MAT, sella_fd_matvec, fd, hP, hPt, rP > 0 ? pevals : nullptr, pscale, m, start.data(), nv0,
                            S->p.gamma, S->p.dav_method, maxiter, nullptr, 0.99, lams.data(), V.get(), AV.get(), &k, &nmv);
    S->neval += sella_fd_calls(fd) * (S->p.threepoint ? 2 : 1);
    if (st != SELLA_OK) { sella_fd_destroy(fd); return cleanup(st); }
    const int kp = sella_fd_npairs(fd);
    std::vector<double> Vs((size_t)n * kp), AVs((size_t)n * kp);
    st = kp > 0 ? sella_fd_pairs(fd, Vs.data(), AVs.data()) : SELLA_OK;
    sella_fd_destroy(fd);
    cleanup(SELLA_OK);
    if (st != SELLA_OK) return st;
    if (kp == 0) { S->first_diag = false; return SELLA_OK; }
    // Ritz rotation of the collected iterates (peswrapper.py:545-551): Atilde = Vs^T symmetrize_Y(Vs, AVs, 2)
    std::vector<double> Yt((size_t)n * kp), At((size_t)kp * kp, 0.0), w(kp), X((size_t)kp * kp), work(kp);
    if (kp == 1) Yt = AVs;
    else SCHK(sella_symmetrize_y(c, Vs.data(), AVs.data(), n, kp, 2, Yt.data()));
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < kp; ++a) {
            const double va = Vs[(size_t)i * kp + a];
            for (int b = 0; b < kp; ++b) At[(size_t)a * kp + b] += va * Yt[(size_t)i * kp + b];
        }
    if (small::sym_eig(kp, At.data(), kp, w.data(), X.data(), kp, work.data()) != 0) {
        set_error("search: Ritz eigenproblem did not converge");
        return SELLA_E_NOCONV;
    }
    std::vector<double> Sm((size_t)n * kp, 0.0), Ym((size_t)n * kp, 0.0);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < kp; ++a) {
            const double va = Vs[(size_t)i * kp + a], ya = AVs[(size_t)i * kp + a];
            for (int b = 0; b < kp; ++b) {
                Sm[(size_t)i * kp + b] += va * X[(size_t)a * kp + b];
                Ym[(size_t)i * kp + b] += ya * X[(size_t)a * kp + b];
            }
        }
    const int ub = update_block(S, Sm.data(), Ym.data(), kp);
    S->first_diag = false;
    if (ub == SELLA_E_UNSUPPORTED) {            // (update_block checks its capacity before it touches anything)
        S->pendS.swap(Sm);
        S->pendY.swap(Ym);
        S->pend_k = kp;
    }
    return ub;
}

// room for one more quasi-Newton pair (and the bookkeeping rows of the one-call step) in the structured forms
bool step_fits(const sella_search* S) {
    const bool pins = !S->idx.empty();
    if (S->r + 4 > S->rank_limit || S->r + 4 > S->cap) return false;
    return !(pins && S->have_view && (S->r_sub + 4 > S->rank_limit_sub || S->r_sub + 4 > S->cap_sub));
}

// optimize.py:363-378
bool wants_diagonalisation(const sella_search* S) {
    if (S->p.diag_every_n >= 0 && S->nsteps_since_diag >= S->p.diag_every_n) return true;
    if (!(S->p.eig && S->nsteps_since_diag >= S->p.nsteps_per_diag)) return false;
    if (S->H_none) return true;
    std::vector<double> pool(S->mu.begin(), S->mu.begin() + S->r);
    const int ncl = std::min(S->p.order, S->n - S->r);
    for (int q = 0; q < ncl; ++q) pool.push_back(S->lam0);
    std::sort(pool.begin(), pool.end());
    for (int q = 0; q < S->p.order && q < (int)pool.size(); ++q)
        if (pool[q] > 0.0) return true;
    return false;
}

int call_opt_step(sella_search* S, int flags, double f_old, bool with_calc = false) {
    sella_opt_step_t& a = S->io;
    const bool pins = !S->idx.empty();
    if ((flags & SELLA_OPT_LEARN) || pins) SCHK(ensure_view(S));
    if (S->r + 4 > S->rank_limit || S->r + 4 > S->cap || (pins && (S->r_sub + 4 > S->rank_limit_sub || S->r_sub + 4 > S->cap_sub))) {
        set_error("search: explicit rank leaves the structured form (%d of %d)", S->r, S->rank_limit);
        return SELLA_E_UNSUPPORTED;
    }
    a.flags = flags;
    a.n = S->n;
    a.B = S->B; a.Wt = S->Wt; a.r = &S->r; a.mu = S->mu.data(); a.lam0 = S->lam0;
    a.update_method = S->p.update_method; a.symm = S->p.symm;
    a.B_stale = S->B_stale; a.Bsub_stale = S->Bsub_stale;
    if (pins) {
        a.Bsub = S->Bsub; a.Wt_sub = S->Wt_sub; a.r_sub = &S->r_sub; a.mu_sub = S->mu_sub.data(); a.idx = S->idx.data(); a.m = S->m;
    } else {
        a.Bsub = SELLA_NO_MAT; a.Wt_sub = SELLA_NO_MAT; a.r_sub = nullptr; a.mu_sub = nullptr; a.idx = nullptr; a.m = 0;
    }
    a.dx = S->dx.data(); a.g_old = S->gold.data(); a.g_new = S->g.data();
    a.f_old = f_old; a.f_new = S->f; a.smag = S->smag;
    a.delta = S->delta; a.rho = S->rho;
    a.delta_min = S->p.delta_min; a.sigma_inc = S->p.sigma_inc; a.sigma_dec = S->p.sigma_dec;
    a.rho_inc = S->p.rho_inc; a.rho_dec = S->p.rho_dec;
    a.stepper_kind = S->p.stepper_kind; a.order = S->p.order; a.cons = S->p.cons; a.maxiter = 1000;
    a.tol = S->p.stepper_kind == SELLA_STEP_QN ? 1e-10 : 1e-15;
    a.s_out = S->s.data();
    if (with_calc) {
        // the force call at S->x rides in front of the update (optstep.hip: one wait for both where the fast form applies)
        bool called = false;
        const int status = opt_step_with_calc(S->c, &a, S->calc, S->x.data(), S->g.data(), &S->f, &called);
        if (called) {                                        // counted and kept whatever became of the step behind it
            ++S->neval;
            S->have_fg = true;
        }
        SCHK(status);
    } else {
        SCHK(sella_opt_step(S->c, &a));
    }
    S->B_stale = a.B_stale;
    S->Bsub_stale = a.Bsub_stale;
    if (flags & SELLA_OPT_LEARN) { S->delta = a.delta; S->rho = a.rho; ++S->nfused; }
    if (flags & SELLA_OPT_PROPOSE) { S->smag = a.smag_out; S->have_step = true; }
    return SELLA_OK;
}

// Sella.step (optimize.py:359-440)
// Stages of one optimizer step for the barriers of a cohort (cohort.h): members of a cohort close up at every stage.
enum { ST_TOP = 0, ST_FIRST_DIAG = 1, ST_PROPOSE = 2, ST_KICK = 3, ST_REDIAG = 4 };

int one_step(sella_search* S) {
    cohort_set_phase(S->c, S->nsteps + 1, ST_TOP);
    cohort_barrier(S->c);
    if (!S->initialized) {                                   // optimize.py:318-326
        if (S->p.eig) {
            cohort_set_phase(S->c, S->nsteps + 1, ST_FIRST_DIAG);
            const int st = diagonalise(S);
            if (st == SELLA_E_UNSUPPORTED && S->pend_k > 0) {    // done, its update pending with the caller; no step taken
                S->nsteps_since_diag = -1;
                S->initialized = true;
            }
            SCHK(st);
            S->nsteps_since_diag = -1;
        }
        S->initialized = true;
    }
    if (S->H_none) {
        // no curvature information yet (eig = False, the default of a minimisation, optimize.py:20-39): the step family
        // sees the identity (stepper.py:62-64) — in structured terms no explicit pairs and lam0 = 1 — and the first
        // secant pair initialises the Hessian (linalg.py:274-289)
        const double f_old = S->f;
        S->gold = S->g;
        S->lam0 = 1.0;
        S->r = 0;
        S->mu.assign(8, 0.0);
        {
            sella_opt_step_t& a = S->io;
            a = sella_opt_step_t();
            a.flags = SELLA_OPT_PROPOSE;
            a.n = S->n;
            a.B = SELLA_NO_MAT; a.Wt = SELLA_NO_MAT; a.r = &S->r; a.mu = S->mu.data(); a.lam0 = 1.0;
            a.update_method = S->p.update_method; a.symm = S->p.symm;
            a.Bsub = SELLA_NO_MAT; a.Wt_sub = SELLA_NO_MAT;
            std::vector<double> gfree;
            int rsub = 0;
            if (!S->idx.empty()) { a.idx = S->idx.data(); a.m = S->m; a.r_sub = &rsub; a.mu_sub = S->mu.data(); }
            a.g_new = S->g.data();
            a.delta = S->delta; a.rho = S->rho;
            a.stepper_kind = S->p.stepper_kind; a.order = S->p.order; a.cons = S->p.cons; a.maxiter = 1000;
            a.tol = S->p.stepper_kind == SELLA_STEP_QN ? 1e-10 : 1e-15;
            a.s_out = S->s.data();
            SCHK(sella_opt_step(S->c, &a));
            S->smag = a.smag_out;
        }
        const bool rediag0 = wants_diagonalisation(S);
        S->nsteps_since_diag = rediag0 ? 0 : S->nsteps_since_diag + 1;
        double gd = 0.0, dd = 0.0;
        for (int i = 0; i < S->n; ++i) {
            S->target[i] = S->x[i] + S->s[i];
            S->dx[i] = S->target[i] - S->x[i];
            gd += S->gold[i] * S->dx[i];
            dd += S->dx[i] * S->dx[i];
        }
        S->x = S->target;
        SCHK(evaluate(S));
        // peswrapper.py:578-602 with H = identity: predicted change g.dx + dx.dx / 2
        const double predicted = gd + 0.5 * dd;
        if (std::sqrt(dd) >= 1e-8) {
            std::vector<double> dg(S->n);
            for (int i = 0; i < S->n; ++i) dg[i] = S->g[i] - S->gold[i];
            SCHK(update_block(S, S->dx.data(), dg.data(), 1));
        }
        if (std::fabs(predicted) >= 1e-14) {                     // optimize.py:413-434 (independent of the diagonalisation)
            const double rho = (S->f - f_old) / predicted;
            if (!(1.0 / S->p.rho_dec <= rho && rho <= S->p.rho_dec)) S->delta = std::max(S->smag * S->p.sigma_dec, S->p.delta_min);
            else if (1.0 / S->p.rho_inc < rho && rho < S->p.rho_inc) S->delta = std::max(S->p.sigma_inc * S->smag, S->delta);
            S->rho = rho;
        } else {
            S->rho = 1.0;
        }
        S->have_step = false;
        if (rediag0) {
            cohort_set_phase(S->c, S->nsteps + 1, ST_REDIAG);
            cohort_barrier(S->c);
            const int st = diagonalise(S);
            if (st == SELLA_E_UNSUPPORTED && S->pend_k > 0) S->count_step_on_exit = true;
            SCHK(st);
        }
        return SELLA_OK;
    }
    // Capacity for THIS step is checked before the geometry moves and the force call is spent: a proposal kept from the
    // previous call (have_step) was made when the explicit rank was smaller by the pair learnt since.
    if (!step_fits(S)) {
        set_error("search: explicit rank leaves the structured form (%d of %d)", S->r, S->rank_limit);
        return SELLA_E_UNSUPPORTED;
    }
    cohort_set_phase(S->c, S->nsteps + 1, ST_PROPOSE);
    if (!S->have_step) SCHK(call_opt_step(S, SELLA_OPT_PROPOSE, S->f));
    const bool rediag = wants_diagonalisation(S);
    S->nsteps_since_diag = rediag ? 0 : S->nsteps_since_diag + 1;
    // PES.kick (peswrapper.py:578-602): move, force call, then everything else in one library call
    const double f_old = S->f;
    S->gold = S->g;
    for (int i = 0; i < S->n; ++i) {
        S->target[i] = S->x[i] + S->s[i];
        S->dx[i] = S->target[i] - S->x[i];
    }
    S->x = S->target;
    S->have_step = false;
    cohort_set_phase(S->c, S->nsteps + 1, ST_KICK);
    cohort_barrier(S->c);
    SCHK(call_opt_step(S, SELLA_OPT_LEARN | (rediag ? 0 : SELLA_OPT_PROPOSE), f_old, true));
    if (rediag) {
        cohort_set_phase(S->c, S->nsteps + 1, ST_REDIAG);
        cohort_barrier(S->c);
        const int st = diagonalise(S);
        if (st == SELLA_E_UNSUPPORTED && S->pend_k > 0) S->count_step_on_exit = true;    // moved, learnt, diagonalised
        SCHK(st);
    }
    return SELLA_OK;
}

}  // namespace

extern "C" sella_ctx* sella_search_ctx(sella_search* S) { return S ? S->c : nullptr; }

extern "C" int sella_search_create(sella_ctx* c, sella_calc* calc, int n, const double* x0, const int* idx, int m,
                                   const sella_search_params_t* p, sella_search** out) {
    if (!c || !calc || !x0 || !p || !out || n <= 0 || n % 3 != 0 || sella_calc_dim(calc) != n || (idx && (m <= 0 || m > n))) {
        set_error("search: invalid arguments");
        return SELLA_E_INVALID;
    }
    if (p->update_method != SELLA_UPD_TS_BFGS || p->cons < 0 || p->cons > 1 || p->stepper_kind < SELLA_STEP_QN ||
        p->stepper_kind > SELLA_STEP_PRFO) {
        set_error("search: configuration outside the library loop (TS-BFGS, trust region / per-atom measure, built-in families)");
        return SELLA_E_UNSUPPORTED;
    }
    sella_search* S = new sella_search();
    S->c = c; S->calc = calc; S->n = n; S->p = *p;
    if (idx) S->idx.assign(idx, idx + m);
    S->m = idx ? m : n;
    S->x.assign(x0, x0 + n);
    S->g.assign(n, 0.0); S->gold.assign(n, 0.0); S->s.assign(n, 0.0); S->dx.assign(n, 0.0); S->target.assign(n, 0.0);
    S->delta = p->delta0;
    S->rho = 1.0;
    S->rank_limit = rank_limit_of(n);
    S->rank_limit_sub = rank_limit_of(S->m);
    S->cap = std::min(n, S->rank_limit + 72);
    S->cap_sub = std::min(S->m, S->rank_limit_sub + 72);
    *out = S;
    return SELLA_OK;
}

// Optimizer.irun: convergence first, then steps until converged or `steps` taken
extern "C" int sella_search_run(sella_search* S, double fmax, long steps, int* converged) {
    if (!S || !converged) return SELLA_E_INVALID;
    if (S->released) { set_error("search: its Hessian was handed over to the caller"); return SELLA_E_INVALID; }
    *converged = 0;
    if (!S->have_fg) SCHK(evaluate(S));
    if (fmax_now(S) < fmax) { *converged = 1; return SELLA_OK; }
    for (long it = 0; it < steps; ++it) {
        const int st = one_step(S);
        if (st != SELLA_OK) {
            if (S->count_step_on_exit) { ++S->nsteps; S->count_step_on_exit = false; }
            return st;
        }
        ++S->nsteps;
        if (fmax_now(S) < fmax) { *converged = 1; return SELLA_OK; }
    }
    return SELLA_OK;
}

// energy and gradient at the current point, if the caller has them already (counted as one force call)
extern "C" int sella_search_seed(sella_search* S, double f, const double* g) {
    if (!S || !g) return SELLA_E_INVALID;
    S->f = f;
    S->g.assign(g, g + S->n);
    S->have_fg = true;
    ++S->neval;
    return SELLA_OK;
}

extern "C" int sella_search_state(sella_search* S, double* x, double* g, double* scalars, long* counters) {
    if (!S) return SELLA_E_INVALID;
    if (x) std::copy(S->x.begin(), S->x.end(), x);
    if (g) std::copy(S->g.begin(), S->g.end(), g);
    if (scalars) {
        scalars[0] = S->f;
        scalars[1] = S->have_fg ? fmax_now(S) : 0.0;
        scalars[2] = S->delta;
        scalars[3] = S->rho;
        double lam = S->H_none ? NAN : S->lam0;              // lowest eigenvalue of the approximate Hessian
        if (!S->H_none && S->r > 0 && (S->r == S->n || S->mu[0] < lam)) lam = S->mu[0];
        scalars[4] = lam;
    }
    if (counters) {
        counters[0] = S->nsteps;
        counters[1] = S->neval;
        counters[2] = S->nfused;
        counters[3] = S->r;
        counters[4] = S->have_view ? S->r_sub : -1;
        counters[5] = S->initialized ? 1 : 0;
    }
    return SELLA_OK;
}

// Hand the approximate Hessian over to the caller (who continues the search with the general driver): the matrix handles
// change owner — the search keeps none and cannot be run again.  mats[4] = B, Wt, Bsub, Wt_sub (SELLA_NO_MAT where absent);
// ints[8] = r, r_sub, rows of Wt, rows of Wt_sub, B_stale, Bsub_stale, steps since the last diagonalisation, first_diag;
// mu / mu_sub: at least `rows` entries each (may be NULL when the search has no Hessian / no view); *lam0.
// Secant pairs of a diagonalisation that no longer fitted the structured form (SELLA_E_UNSUPPORTED from sella_search_run):
// *k pairs, Sm / Ym (n x k) row-major (NULL: only the count).  The caller applies them as ONE block update
// (ApproximateHessian.update, linalg.py:274-304) — then its state is what the reference's would be after PES.diag.
extern "C" int sella_search_pending_pairs(sella_search* S, int* k, double* Sm, double* Ym) {
    if (!S || !k) return SELLA_E_INVALID;
    *k = S->pend_k;
    if (S->pend_k > 0 && Sm && Ym) {
        memcpy(Sm, S->pendS.data(), S->pendS.size() * sizeof(double));
        memcpy(Ym, S->pendY.data(), S->pendY.size() * sizeof(double));
    }
    return SELLA_OK;
}

extern "C" int sella_search_release_hessian(sella_search* S, sella_mat* mats, long* ints, double* mu, double* mu_sub,
                                            double* lam0) {
    if (!S || !mats || !ints || !lam0) return SELLA_E_INVALID;
    mats[0] = S->B; mats[1] = S->Wt; mats[2] = S->Bsub; mats[3] = S->Wt_sub;
    ints[0] = S->H_none ? -1 : S->r;
    ints[1] = S->have_view ? S->r_sub : -1;
    ints[2] = S->H_none ? 0 : S->cap;
    ints[3] = S->have_view ? S->cap_sub : 0;
    ints[4] = S->B_stale;
    ints[5] = S->Bsub_stale;
    ints[6] = S->nsteps_since_diag;
    ints[7] = S->first_diag ? 1 : 0;
    *lam0 = S->lam0;
    if (!S->H_none && mu) std::copy(S->mu.begin(), S->mu.begin() + std::min((size_t)S->cap, S->mu.size()), mu);
    if (S->have_view && mu_sub) std::copy(S->mu_sub.begin(), S->mu_sub.begin() + std::min((size_t)S->cap_sub, S->mu_sub.size()), mu_sub);
    S->B = S->Wt = S->Bsub = S->Wt_sub = SELLA_NO_MAT;
    S->H_none = true;
    S->have_view = false;
    S->released = true;
    return SELLA_OK;
}

extern "C" int sella_search_destroy(sella_search* S) {
    if (!S) return SELLA_OK;
    for (sella_mat h : {S->B, S->Wt, S->Bsub, S->Wt_sub})
        if (h != SELLA_NO_MAT) sella_mat_free(S->c, h);
    delete S;
    return SELLA_OK;
}
