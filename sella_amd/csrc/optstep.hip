// One optimizer step of the saddle-point search as ONE library call (sella/optimize/optimize.py:359-440 with
// peswrapper.py:578-602 and linalg.py:274-304 behind it): what `Sella.step` does between two force calls —
//
//   learn    predicted change of the quadratic model along the step taken, ratio of actual to predicted change,
//            quasi-Newton update of the approximate Hessian with the new secant pair (matrix, structured
//            eigendecomposition, principal-submatrix view of pinned-coordinate constraints);
//   adapt    trust radius from the ratio (optimize.py:413-434);
//   propose  step family in the eigenbasis + restricted-step root find at the new point (restricted_step.py:78-120)
//
// — without returning to the host language in between.  The calculator boundary (energy + gradient of the new geometry,
// peswrapper.py:413-418) stays with the caller: it hands over f and g, and receives the next step.  Each phase calls the
// same routines the one-phase entry points call (sella_update_h_lr, sella_stepper_create_lr, sella_restricted_step), so
// the results are theirs bit for bit; what goes away is the interpreter between ~300 fine-grained calls, and with it
// the reason host threads could not share a GPU (the call releases the interpreter lock for its whole duration).
#include "internal.h"

#include <cmath>
#include <limits>

using namespace sella;

namespace {

// class attributes of the step families (sella/optimize/stepper.py:20-41, 66-70, 114-120)
struct Family {
    double alpha0, alphamin, alphamax, slope;
    int newton_safe;
};

Family family_of(int kind) {
    if (kind == SELLA_STEP_QN) return {0.0, 0.0, std::numeric_limits<double>::infinity(), -1.0, 1};
    return {1.0, 0.0, 1.0, 1.0, 0};                        // rfo, prfo
}

}  // namespace

static int opt_step_general(sella_ctx* c, sella_opt_step_t* a);

extern "C" int sella_opt_step(sella_ctx* c, sella_opt_step_t* a) {
    if (!c || !a || a->n <= 0 || !a->r || !a->mu) return SELLA_E_INVALID;
    a->ratio_valid = 0;
    a->updated = 0;
    a->nrank1 = a->nrank1_sub = 0;
    a->nalpha = 0;
    {
        // the fast form: structured decompositions updated in coordinates, every decision on the device (lrstep.hip)
        bool handled = false;
        SCHK(lr_fused_step(c, a, &handled));
        if (handled) return SELLA_OK;
    }
    return opt_step_general(c, a);
}

// The same with the force call at the new geometry INSIDE (csrc/search.hip): where the fast form applies, the calculator's
// kernels are queued in front of the update that consumes their gradient and the host waits once for both; otherwise
// the force call is made here and the step proceeds as sella_opt_step.  g_new (n) and *f_new receive gradient and energy.
int sella::opt_step_with_calc(sella_ctx* c, sella_opt_step_t* a, sella_calc* calc, const double* x, double* g_new,
                              double* f_new, bool* force_call_made) {
    if (force_call_made) *force_call_made = false;
    if (!c || !a || !calc || !x || !g_new || !f_new || a->n <= 0 || !a->r || !a->mu) return SELLA_E_INVALID;
    a->ratio_valid = 0;
    a->updated = 0;
    a->nrank1 = a->nrank1_sub = 0;
    a->nalpha = 0;
    a->g_new = g_new;
    CalcPipe pipe;
    pipe.calc = calc; pipe.x = x; pipe.g_out = g_new;
    bool handled = false;
    // (the status of the step is propagated only after the force call has been accounted for: a step that fails behind a
    // finished force call must leave the caller with energy, gradient and call count of the geometry it moved to)
    int status = lr_fused_step(c, a, &handled, &pipe);
    if (pipe.done) {
        *f_new = pipe.f;
        if (force_call_made) *force_call_made = true;
        if (status != SELLA_OK || handled) return status;
    } else {
        SCHK(status);
        SCHK(sella_calc_eval(calc, x, f_new, g_new));
        if (force_call_made) *force_call_made = true;
        a->f_new = *f_new;
        SCHK(lr_fused_step(c, a, &handled));
        if (handled) return SELLA_OK;
    }
    return opt_step_general(c, a);
}

static int opt_step_general(sella_ctx* c, sella_opt_step_t* a) {
    const int n = a->n;
    const bool view = a->idx != nullptr && a->m > 0;
    // general route: the one-phase entry points in sequence.  They update the dense matrices in place: rebuild them
    // first if they lag behind the decompositions.
    if (a->flags & SELLA_OPT_LEARN) {
        if (a->B_stale) {
            SCHK(sella_lr_materialize(c, a->B, a->Wt, *a->r, a->mu, a->lam0));
            a->B_stale = 0;
        }
        if (view && a->Bsub_stale && a->r_sub) {
            SCHK(sella_lr_materialize(c, a->Bsub, a->Wt_sub, *a->r_sub, a->mu_sub, a->lam0));
            a->Bsub_stale = 0;
        }
    }
    if (a->flags & SELLA_OPT_LEARN) {
        if (!a->dx || !a->g_old || !a->g_new) return SELLA_E_INVALID;
        // quadratic model along the step (peswrapper.py:446-449, 458-462): g.dx + dx.(B dx) / 2
        std::vector<double>& Bdx = c->hbuf_a;
        Bdx.resize((size_t)n);
        SCHK(sella_symm_mm(c, a->B, a->dx, 1, Bdx.data()));
        double gd = 0.0, dBd = 0.0, dd = 0.0;
        for (int i = 0; i < n; ++i) {
            gd += a->g_old[i] * a->dx[i];
            dBd += a->dx[i] * Bdx[i];
            dd += a->dx[i] * a->dx[i];
        }
        const double predicted = gd + 0.5 * dBd;
        a->df_pred = predicted;
        if (std::fabs(predicted) >= 1e-14) {
            a->ratio = (a->f_new - a->f_old) / predicted;
            a->ratio_valid = 1;
        }
        // linalg.py:274-304; a step shorter than 1e-8 leaves B alone (hessian_update.py:48-49)
        if (std::sqrt(dd) >= 1e-8) {
            std::vector<double>& dg = c->hbuf_b;
            dg.resize((size_t)n);
            for (int i = 0; i < n; ++i) dg[i] = a->g_new[i] - a->g_old[i];
            SCHK(sella_update_h_lr(c, a->B, a->Wt, a->r, a->mu, a->lam0, a->dx, dg.data(), n, 1, a->update_method,
                                   a->symm, &a->nrank1, view ? a->Bsub : SELLA_NO_MAT,
                                   (view && a->r_sub) ? a->Wt_sub : SELLA_NO_MAT, a->r_sub, a->mu_sub, a->idx,
                                   view ? a->m : 0, &a->nrank1_sub));
            a->updated = 1;
        }
        // optimize.py:413-434
        if (!a->ratio_valid) {
            a->rho = 1.0;
        } else {
            const double rho = a->ratio;
            if (!(1.0 / a->rho_dec <= rho && rho <= a->rho_dec))
                a->delta = std::fmax(a->smag * a->sigma_dec, a->delta_min);
            else if (1.0 / a->rho_inc < rho && rho < a->rho_inc)
                a->delta = std::fmax(a->sigma_inc * a->smag, a->delta);
            a->rho = rho;
        }
    }
    if (a->flags & SELLA_OPT_PROPOSE) {
        if (!a->g_new || !a->s_out) return SELLA_E_INVALID;
        const Family fam = family_of(a->stepper_kind);
        sella_stepper* st = nullptr;
        int rc;
        if (view) {
            if (!a->r_sub || !a->mu_sub) return SELLA_E_INVALID;
            std::vector<double>& gs = c->hbuf_b;
            gs.resize((size_t)a->m);
            for (int j = 0; j < a->m; ++j) gs[j] = a->g_new[a->idx[j]];
            SCHK(sella_stepper_create_lr(c, a->stepper_kind, a->Wt_sub, *a->r_sub, a->mu_sub, a->lam0, gs.data(), a->m,
                                         a->order, &st));
            stepper_set_fast_search(st, c->opt.rs_fast != 0);
            rc = sella_restricted_step(st, a->cons, a->delta, nullptr, nullptr, nullptr, fam.alpha0, fam.alphamin,
                                       fam.alphamax, fam.slope, fam.newton_safe, 1, a->tol, a->maxiter, a->idx, n,
                                       a->s_out, &a->smag_out, nullptr, &a->nalpha);
        } else {
            SCHK(sella_stepper_create_lr(c, a->stepper_kind, a->Wt, *a->r, a->mu, a->lam0, a->g_new, n, a->order, &st));
            stepper_set_fast_search(st, c->opt.rs_fast != 0);
            rc = sella_restricted_step(st, a->cons, a->delta, nullptr, nullptr, nullptr, fam.alpha0, fam.alphamin,
                                       fam.alphamax, fam.slope, fam.newton_safe, 1, a->tol, a->maxiter, nullptr, 0,
                                       a->s_out, &a->smag_out, nullptr, &a->nalpha);
        }
        sella_stepper_destroy(st);
        if (rc != SELLA_OK) return rc;
    }
    return SELLA_OK;
}
