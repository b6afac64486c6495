/*
 * sella_hip.h — C ABI of libsella_hip.so, the MI355X (gfx950) implementation of the
 * inner saddle-point linear-algebra loop of Sella.
 *
 * This is the drop-in boundary.  The reference has no FFI; the seam it
 * offers for exactly this path is sella/_gpu.py (numpy-in / numpy-out helpers backed by
 * torch.cuda) plus the function-level seams one level up.  Each entry point below names
 * the reference interface it replaces (path:line under /root/reference).  INTEGRATION.md
 * shows the ctypes stub a Sella maintainer would add.
 *
 * Conventions
 *   - all matrices are fp64, C-contiguous row-major; host pointers are borrowed for the
 *     duration of the call only and are never written unless documented as outputs;
 *   - device memory is owned by the context and addressed through integer handles;
 *   - every function returns 0 on success or a negative SELLA_E_* code, the message is
 *     available from sella_last_error(); no exceptions cross the boundary;
 *   - one host thread per context, one HIP stream per context; calls are synchronous at
 *     the boundary (outputs are valid on return);
 *   - there is NO CPU fallback: without a usable HIP device sella_ctx_create fails.
 */
#ifndef SELLA_HIP_H
#define SELLA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sella_ctx sella_ctx;
typedef int sella_mat;            /* handle of a device-resident row-major matrix */
#define SELLA_NO_MAT (-1)

enum {
    SELLA_OK = 0,
    SELLA_E_INVALID = -1,         /* bad argument / shape mismatch                       */
    SELLA_E_HIP = -2,             /* HIP runtime error (message has the HIP string)      */
    SELLA_E_NOMEM = -3,           /* device allocation failed (cf. _gpu.py:44-52 OOM)    */
    SELLA_E_NOCONV = -4,          /* iteration limit hit (MGS sweeps, secular solver..)  */
    SELLA_E_CALLBACK = -5,        /* host matvec callback reported failure               */
    SELLA_E_NODEVICE = -6,        /* no HIP device visible                               */
    SELLA_E_UNSUPPORTED = -7
};

const char* sella_last_error(void);
const char* sella_version(void);

/* ---- device / context ------------------------------------------------------------ */
/* replaces the implicit torch.cuda device + availability probe, sella/_gpu.py:17-41    */
int sella_device_count(int* count);
int sella_ctx_create(int device, sella_ctx** ctx);
int sella_ctx_destroy(sella_ctx* ctx);
int sella_ctx_sync(sella_ctx* ctx);
int sella_ctx_device_name(sella_ctx* ctx, char* buf, int buflen);
/* integer tuning knobs (kernel variant selection for benchmarking); unknown key -> error.  Keys (defaults):
 *   gemv_rw (0 = by size) rows per workgroup of the streaming matvec | gemm_mfma (1) GEMMs on the matrix cores |
 *   gemm_tile128 (1) 128x128 GEMM tiles for large products | panel_mfma (1), panel_rows (0 = by size) the
 *   H.V block product | host_scalars (0) zero-copy scalars | rank2k_stream (1) mirror-free trailing update |
 *   eigh_nb (16) panel width, eigh_leaf (16) leaf size, eigh_wy_mfma (1) MFMA back-transformation |
 *   eigh_symv_min (5120) trailing blocks of the tridiagonalisation with at least this many rows take the
 *   symmetric-aware matvec (upper triangle only, fixed-order partial sums; 0: never), eigh_symv_tr (64) rows per tile
 *   of it, eigh_symv_tri (1) trailing update on the upper triangle only while it runs | eigh_wy_nb64_min (2560) 64
 *   instead of 32 reflectors per block of the back-transformation from this many rows on |
 *   dav_fuse_scale (1), dav_zero_copy (0) Davidson chain: diagonal scaling inside the residual kernel, coefficients read
 *   from pinned host memory |
 *   eigh_tail_lds (128) the last <= 128 columns of the tridiagonalisation inside one workgroup, the block in LDS |
 *   rs_batch (1) bisection phase of sella_restricted_step: 15 trial alphas per device round trip |
 *   panel_small (2048) panel products with <= 64 rows and <= 16 right-hand sides take the split-K kernels from this
 *   many columns on | bd_pipeline (1) sella_davidson_block as a pipelined iteration: projection and block Gram
 *   matrix from one panel product, two polled waits per iteration (0: the general loop) | bd_early_matvec (1) ... with A applied
 *   to the raw correction block while the host orthonormalises it (error budget for the transformed A T) | lr_dev (1) sella_opt_step updates structured eigendecompositions in
 *   coordinates with every decision on the device (0: the host-planned rank-one merges of sella_update_h_lr) |
 *   eigh_wy_waves (4), eigh_wy_rows (16) wavefronts / rows of X per workgroup of the back-transformation (8, 16 / 32
 *   measured equal or slower) | rank2k_fixed (1) trailing update with all loads up front | lr_cholqr (1) block of update
 *   vectors by rank-revealing Cholesky-QR |
 *   rs_fast (1) sella_opt_step finds the restricted step by interpolating batches of 15 trial alphas instead of the
 *   reference's Newton / bisection schedule (same root) |
 *   eigh_upd_max (1024): trailing blocks of the tridiagonalisation with at most this many rows take ONE launch per column
 *   (the block kept up to date by the launch itself; 0: the blocked two-launch chain throughout), eigh_upd_rows (0 = 2),
 *   eigh_upd_nt (512) rows / most threads per workgroup of that launch | eigh_gemv_flat (1) trailing matvec of the blocked
 *   chain with every load issued before the first wait (0: the loop form; same sums, bit for bit), eigh_dc_pipeline (1)
 *   divide & conquer with ONE host wait per level (0: two; same results bit for bit), eigh_wy_overlap (1) compact-WY
 *   factors on a second stream beside divide & conquer | h2d_kernel_min (16384) host-to-device payloads of at least this
 *   many bytes are copied by a kernel that reads the pinned staging ring (0: always the runtime's copy) |
 *   lr_chain (1) the structured quasi-Newton update of sella_opt_step as the fused launch chain of round 4 (0: round 3's
 *   kernels), lr_pipe (1) the force call queued in front of the update that consumes it, rs_batch_result (1) final step
 *   read from the batch of trial alphas that produced it, lr_overlap (0) view job on a second stream |
 *   emt_hcap (8) neighbour-list slots per thread of the EMT kernels (1 .. 8; tests lower it to reach the overflow path). */
int sella_ctx_set_option(sella_ctx* ctx, const char* key, long value);

/* ---- device matrices --------------------------------------------------------------- */
/* to_gpu(A): sella/_gpu.py:55-67 (upload, cached by ApproximateHessian linalg.py:197-207) */
int sella_mat_upload(sella_ctx* ctx, const double* A, int rows, int cols, sella_mat* h);
int sella_mat_alloc(sella_ctx* ctx, int rows, int cols, sella_mat* h);
int sella_mat_set(sella_ctx* ctx, sella_mat h, const double* A);        /* overwrite    */
int sella_mat_download(sella_ctx* ctx, sella_mat h, double* out);       /* .cpu().numpy() */
int sella_mat_shape(sella_ctx* ctx, sella_mat h, int* rows, int* cols);
int sella_mat_copy(sella_ctx* ctx, sella_mat src, sella_mat* dst);
int sella_mat_transpose(sella_ctx* ctx, sella_mat src, sella_mat* dst);
/* new matrix holding rows [row0, row0 + nrows) of src (the explicit eigenvectors of a structured eigendecomposition
 * are the leading rows of a larger buffer)                                                                        */
int sella_mat_rows(sella_ctx* ctx, sella_mat src, int row0, int nrows, sella_mat* dst);
/* dst rows [0, nrows) <- src rows [0, nrows) (equal column counts): growing such a buffer                          */
int sella_mat_copy_into(sella_ctx* ctx, sella_mat src, sella_mat dst, int nrows);
/* A[i][i] += alpha: B = lam0 * I of the first-update rule (sella/linalg.py:274-289) is a zero matrix plus this      */
int sella_mat_add_diag(sella_ctx* ctx, sella_mat h, double alpha);
int sella_mat_free(sella_ctx* ctx, sella_mat h);
/* C = alpha * A + beta * B (same shapes; B may be SELLA_NO_MAT with beta ignored)       */
int sella_mat_axpby(sella_ctx* ctx, double alpha, sella_mat A, double beta, sella_mat B,
                    sella_mat* C);

/* ---- dense products ------------------------------------------------------------------ */
/* Y = A X, A (n x n) resident, X and Y host (n x k row-major).
 * ApproximateHessian._matvec/_matmat  sella/linalg.py:324-335; dense A.dot(t) in
 * sella/eigensolvers.py:52,112; B@S in sella/hessian_update.py:119,160-176.             */
int sella_symm_mm(sella_ctx* ctx, sella_mat A, const double* X, int k, double* Y);
/* Y = A^T X for a general resident A (rows x cols), X host (rows x k), Y (cols x k)     */
int sella_gemm_tn_host(sella_ctx* ctx, sella_mat A, const double* X, int k, double* Y);
/* out (m x m) = U^T H U, H (n x n) resident, U host (n x m).
 * gpu_project  sella/_gpu.py:114-132 (peswrapper.py:374,382; linalg.py:306-317)         */
int sella_project(sella_ctx* ctx, sella_mat H, const double* U, int m, double* out);
/* same, result stays on the device as a new handle (and optionally on the host)        */
int sella_project_dev(sella_ctx* ctx, sella_mat H, sella_mat U, sella_mat* out);
/* general C = op(A) op(B) on resident matrices (test / building block)                 */
int sella_gemm(sella_ctx* ctx, int transA, int transB, double alpha, sella_mat A,
               sella_mat B, double beta, sella_mat C);

/* ---- symmetric eigensolver ------------------------------------------------------------ */
/* gpu_eigh / gpu_eigh_t  sella/_gpu.py:70-97 (torch.linalg.eigh).  w (n) ascending.
 * V: new handle holding the eigenvectors as COLUMNS (V[i][j] = component i of vector j),
 * Vt: new handle with the eigenvectors as ROWS.  Either pointer may be NULL.             */
int sella_eigh(sella_ctx* ctx, sella_mat A, double* w, sella_mat* V, sella_mat* Vt);

/* Building block of the eigensolver's divide-and-conquer stage, exposed for testing and reuse:
 * eigendecomposition of diag(D) + rho w w^T (D strictly ascending, w_i != 0, rho > 0).
 * lam (K) ascending; Ut (K x K row-major, may be NULL) has the eigenvectors as rows.          */
int sella_rank1_eig(sella_ctx* ctx, int K, const double* D, const double* w, double rho,
                    double* lam, double* Ut);

/* ---- thin QR ---------------------------------------------------------------------------- */
/* gpu_qr(A) economy mode  sella/_gpu.py:100-111 (peswrapper.py:691).  A host (m x n),
 * m >= n; Q (m x n), R (n x n) host outputs.                                             */
int sella_qr_thin(sella_ctx* ctx, const double* A, int m, int n, double* Q, double* R);

/* ---- Gram-Schmidt --------------------------------------------------------------------- */
/* modified_gram_schmidt(X, Y)  sella/utilities/math.pyx:143-159 (mgs :74-140).
 * X host (n x nx), Y host (n x ny) or NULL; out host (n x nx); *kept = columns kept.
 * Returns SELLA_E_NOCONV where the reference raises RuntimeError("MGS failed.").         */
int sella_mgs(sella_ctx* ctx, const double* X, int n, int nx, const double* Y, int ny,
              double eps1, double eps2, int maxiter, double* out, int* kept);

/* ---- Davidson / Rayleigh-Ritz ----------------------------------------------------------- */
/* rayleigh_ritz(A, gamma, P, B=None, v0, vref, vreftol, method, maxiter)
 *   sella/eigensolvers.py:31-112, expand :115-153.
 * A: resident dense matrix, or SELLA_NO_MAT with a callback (the finite-difference
 *    operator NumericalHessian, sella/linalg.py:39-95, lives behind the calculator boundary:
 *    a host-language callback for an ASE calculator, sella_fd_matvec for one of the library's own).
 * P: given by its eigendecomposition P = Q diag(pevals) Q^T with Q resident both as columns
 *    (Pvecs) and as rows (PvecsT) — exactly what sella_eigh returns — or Pvecs = SELLA_NO_MAT
 *    for P = pscale * I.
 * v0: host (n x nv0) start block (nv0 >= 1).  vref may be NULL.
 * Outputs (host): lams (kmax), V and AV (n x k row-major, Ritz vectors as columns), *k.
 * STRUCTURED P: Pvecs (n x r) and PvecsT (r x n) with r < n hold r explicit eigenpairs (pevals, r entries) and the
 *    remaining n - r eigenvalues all equal pscale, their eigenspace being the orthogonal complement of the r vectors:
 *    P = pscale (I - W^T W) + W^T diag(pevals) W — what an approximate Hessian that started as a scaled identity
 *    (sella/linalg.py:274-289) is after any number of quasi-Newton updates.  (P - theta)^-1 then costs O(n r).
 * RE-ENTRANCY CONTRACT of the callbacks (sella_matvec_fn, sella_allgather_fn).  A callback runs on the thread that
 *    called the solver and MAY CALL ANY ENTRY POINT OF THIS LIBRARY ON THE SAME CONTEXT: create, upload, free and
 *    factorise matrices, run sella_eigh / sella_qr_thin / another sella_davidson, synchronise.  (NumericalHessian._matvec,
 *    sella/linalg.py:39-95, evaluates a calculator, and InternalPES transforms the vector through device-resident
 *    Jacobians on the way.)  The library guarantees: (1) matrix handles the solver was given stay valid and keep their
 *    device address while they are live — the handle table never moves an entry; (2) every scratch slot, scalar exchange
 *    buffer and staging buffer of the interrupted call is parked for the duration of the callback, the nested calls work
 *    on a set of their own per call depth; (3) device work queued by the callback is ordered behind the solver's on the
 *    context's single stream.  The callback must NOT free or overwrite the matrices passed to the interrupted solver
 *    (A, Pvecs, PvecsT), must not destroy the context, and must not call the library on this context from another
 *    thread.  v and Av (send / recv) are only valid until it returns.  A non-zero return aborts the solver with
 *    SELLA_E_CALLBACK.                                                                                              */
typedef int (*sella_matvec_fn)(void* user, const double* v, double* Av, int n);
enum { SELLA_DAV_LANCZOS = 0, SELLA_DAV_GD = 1, SELLA_DAV_JD0 = 2, SELLA_DAV_JD0_ALT = 3,
       SELLA_DAV_MJD0 = 4, SELLA_DAV_MJD0_ALT = 5 };
int sella_davidson(sella_ctx* ctx, sella_mat A, sella_matvec_fn matvec, void* user,
                   sella_mat Pvecs, sella_mat PvecsT, const double* pevals, double pscale,
                   int n, const double* v0, int nv0, double gamma, int method, int maxiter,
                   const double* vref, double vreftol,
                   double* lams, double* V, double* AV, int* k, int* nmatvec);

/* Block Davidson for the lowest `nev` eigenpairs of a large dense symmetric operator, `block` (<= 16) new
 * vectors per iteration with the operator streamed once per block on the matrix cores (BASELINE.json
 * configs[4]; the reference adds one vector per iteration, sella/eigensolvers.py:111-112, so there is no
 * trajectory to match: the converged pairs equal those of exact(), sella/eigensolvers.py:9-28).
 * A: resident (n x n), or — with `gather` — this rank's row panel A[row0 : row0 + rows, :] of a matrix whose rows
 *    are split evenly (ceil(n / world) per rank) over `world` ranks; `gather` must all-gather `bytes` bytes from
 *    every rank's device buffer `send` into `recv` (rank-major) on the given HIP stream (ncclAllGather).
 * Preconditioner: the eigendecomposition of an approximate operator P (Pvecs / PvecsT / pevals as for
 *    sella_davidson: t = Q (d - theta)^-1 Q^T r, the 'gd' correction of sella/eigensolvers.py:119-121), else
 *    diag (n, host: t_i = r_i / (diag_i - theta)), else none.
 * V0 (n x nv0, host, nv0 <= 16) start block or NULL.  tol: a pair counts as converged when
 *    |r| <= tol |theta| (the reference's gamma test, sella/eigensolvers.py:80-89).  maxvec: basis limit before a thick
 *    restart, 0 = nev + 3 block (room for two blocks between restarts; at least nev + 2 block).
 * Outputs (host): lams (nev), V (n x nev row-major), res (nev residual norms, may be NULL), *niter,
 *    *nmatvec (operator columns applied), *nconv (pairs converged; nev on success).
 * The gather callback is stream-ordered (no synchronisation around it); the re-entrancy contract above applies.     */
typedef int (*sella_allgather_fn)(void* user, const void* send, void* recv, size_t bytes, void* hip_stream);
int sella_davidson_block(sella_ctx* ctx, sella_mat A, int n, int row0, int world,
                         sella_allgather_fn gather, void* user, sella_mat Pvecs, sella_mat PvecsT,
                         const double* pevals, const double* diag, const double* V0, int nv0, int nev,
                         int block, int maxvec, double tol, int maxiter, double* lams, double* V,
                         double* res, int* niter, int* nmatvec, int* nconv);

/* Raw copy between buffers handed to a callback (kind 0: device -> device, 1: device -> host, 2: host -> device),
 * ordered on the context's stream and complete on return.  Lets a host-side all-gather (the gloo shim of the CPU
 * tests) stage the device buffers of sella_davidson_block; RCCL takes the device pointers directly.              */
int sella_dev_copy(sella_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
/* The context's HIP stream (hipStream_t) and the device address / leading dimension of a resident matrix: the
 * handles a collective library bound from the host language (librccl through ctypes: ncclAllGather(send, recv,
 * count, ncclDouble, comm, stream)) needs to run on the library's own buffers in stream order, without PyTorch.
 * The reference has no collective anywhere; this is new surface for configs[3] / [4].     */
int sella_ctx_stream(sella_ctx* ctx, void** hip_stream);
int sella_mat_ptr(sella_ctx* ctx, sella_mat h, void** device_ptr, int* ld);

/* ---- quasi-Newton update ------------------------------------------------------------------ */
/* update_H(B, S, Y, method, symm, lams, vecs, B_gpu, evals_gpu, evecs_gpu)
 *   sella/hessian_update.py:40-111, formulas :114-157, torch variant :160-203.
 * B resident (n x n) is updated IN PLACE and symmetrised in the same pass.
 * evecs/evals: eigendecomposition of B (columns), needed by TS-BFGS / BFGS_auto only.
 * S, Y host (n x k).  symm in {-1 (None), 0, 1, 2}.                                          */
enum { SELLA_UPD_TS_BFGS = 0, SELLA_UPD_BFGS = 1, SELLA_UPD_PSB = 2, SELLA_UPD_DFP = 3,
       SELLA_UPD_SR1 = 4, SELLA_UPD_GREENSTADT = 5, SELLA_UPD_BFGS_AUTO = 6 };
int sella_update_h(sella_ctx* ctx, sella_mat B, sella_mat evecs, sella_mat evecsT,
                   const double* evals, const double* S, const double* Y, int n, int k,
                   int method, int symm);
/* sella_update_h that also carries the eigendecomposition of B over to the updated matrix: the
 * update B+ - B has rank 2k..4k, so its eigenpairs follow from those of B by that many rank-one
 * modifications (secular equation + one GEMM each, eigh.hip) instead of a new factorisation —
 * what the reference pays torch.linalg.eigh for after every step (linalg.py:174-231).
 * evals (n, in/out), evecs / evecsT updated in place.  If the update has rank > max_rank (or no
 * eigenvectors are given) only B is updated and *nrank1 = -1: the caller then recomputes the
 * eigendecomposition with sella_eigh.  Otherwise *nrank1 = rank-one modifications applied.          */
int sella_update_h_eig(sella_ctx* ctx, sella_mat B, sella_mat evecs, sella_mat evecsT, double* evals,
                       const double* S, const double* Y, int n, int k, int method, int symm,
                       int max_rank, int* nrank1);
/* The same, keeping a principal submatrix of B in step with it: Bsub = B[idx][idx] (idx ascending, m
 * entries) is what `get_HL_projected` (sella/peswrapper.py:363-386) yields when the constraints pin single
 * Cartesian coordinates — U^T B U with U columns of the identity — and the reference re-diagonalises it at
 * every step.  Bsub receives the update restricted to idx; if evecs_sub / evecsT_sub / evals_sub are given
 * (handles != SELLA_NO_MAT) its eigendecomposition is carried the same way, *nrank1_sub as above.          */
int sella_update_h_eig_view(sella_ctx* ctx, sella_mat B, sella_mat evecs, sella_mat evecsT, double* evals,
                            const double* S, const double* Y, int n, int k, int method, int symm,
                            int max_rank, int* nrank1, sella_mat Bsub, sella_mat evecs_sub,
                            sella_mat evecsT_sub, double* evals_sub, const int* idx, int m,
                            int* nrank1_sub);
/* STRUCTURED eigendecomposition.  An approximate Hessian initialised by the first-update rule (sella/linalg.py:274-289:
 * B = lam0 I + update) stays lam0 I + (rank r) for ever: r explicit eigenpairs — mu (host, ascending) and the leading r
 * rows of Wt (capacity rows x n, resident) — plus the eigenvalue lam0 on the orthogonal complement of their span.  The
 * reference re-diagonalises the dense matrix after every update (linalg.py:174-231, torch.linalg.eigh); sella_update_h_eig
 * carries n explicit eigenvectors by rank-one merges with O(n^2) passes; here a rank-one term costs O(n r): the component
 * of its vector outside span(W) joins W as a row with eigenvalue lam0 and the merge runs on r + 1 rows.
 *   B (n x n resident) receives the update as in sella_update_h; *r and mu are updated (r grows by at most 4k per call:
 *   the caller provides capacity); *nrank1 = rank-one merges applied.
 *   Optional principal-submatrix view Bsub = B[idx][idx] with a structured eigendecomposition of its own
 *   (Wt_sub capacity x m, *r_sub, mu_sub; lam0 the same) kept in step; pass Bsub = SELLA_NO_MAT for none, or
 *   Wt_sub = SELLA_NO_MAT to update the view's matrix only.                                                          */
int sella_update_h_lr(sella_ctx* ctx, sella_mat B, sella_mat Wt, int* r, double* mu, double lam0, const double* S,
                      const double* Y, int n, int k, int method, int symm, int* nrank1, sella_mat Bsub,
                      sella_mat Wt_sub, int* r_sub, double* mu_sub, const int* idx, int m, int* nrank1_sub);
/* Structured eigendecomposition of the principal submatrix B[idx][idx] (idx ascending, m entries) from that of B:
 * U^T B U for U = columns of the identity, sella/peswrapper.py:363-386, without an m x m eigh.
 * Wt_sub: capacity (>= min(r, m)) x m; outputs *r_sub, mu_sub (capacity entries).                                    */
int sella_lr_restrict(sella_ctx* ctx, sella_mat Wt, int r, const double* mu, double lam0, const int* idx, int m,
                      sella_mat Wt_sub, int* r_sub, double* mu_sub);
/* symmetrize_Y(S, Y, symm)  sella/hessian_update.py:12-37; out host (n x k)                   */
int sella_symmetrize_y(sella_ctx* ctx, const double* S, const double* Y, int n, int k,
                       int symm, double* out);

/* ---- step solve -------------------------------------------------------------------------- */
/* One evaluation of Stepper.get_s(alpha)  sella/optimize/stepper.py:82-96 (QN), :128-157
 * (RFO), :179-185 (P-RFO) in the eigenbasis of the (projected) Hessian:
 *   evecs (m x m resident, columns) / evecsT (rows), evals (m), g (m) host.
 * kind: 0 = qn, 1 = rfo, 2 = prfo.  Outputs s, dsda (m) host.                               */
enum { SELLA_STEP_QN = 0, SELLA_STEP_RFO = 1, SELLA_STEP_PRFO = 2, SELLA_STEP_QN_IRC = 3 };
typedef struct sella_stepper sella_stepper;
int sella_stepper_create(sella_ctx* ctx, int kind, sella_mat evecs, sella_mat evecsT,
                         const double* evals, const double* g, int m, int order,
                         sella_stepper** st);
/* The same families on a STRUCTURED eigendecomposition (see sella_update_h_lr): r explicit eigenpairs (mu ascending,
 * leading rows of Wt) + eigenvalue lam0 on the complement.  A mode without a gradient component gets a zero step in
 * every family, so the n - r cluster modes are represented by ONE — the normalised component of g outside span(W) —
 * plus min(order, n - r - 1) weightless copies that keep "the `order` lowest modes" and the RFO root index counting the
 * cluster's multiplicity.  The stepper then behaves exactly like one created from the dense eigendecomposition.      */
int sella_stepper_create_lr(sella_ctx* ctx, int kind, sella_mat Wt, int r, const double* mu, double lam0,
                            const double* g, int n, int order, sella_stepper** st);
int sella_stepper_get_s(sella_stepper* st, double alpha, double* s, double* dsda);
int sella_stepper_destroy(sella_stepper* st);
/* QuasiNewtonIRC (sella/optimize/stepper.py:99-111): kind SELLA_STEP_QN_IRC evaluates
 * s(alpha) = -V (V^T g + alpha V^T d1) / (|lam| + alpha); d1hat = V^T d1 (m entries: the accumulated IRC
 * displacement in the eigenbasis of the projected Hessian).  Must be set before the first evaluation.            */
int sella_stepper_set_d1hat(sella_stepper* st, const double* d1hat, int m);

/* The whole restricted-step root find, BaseRestrictedStep.get_s of sella/optimize/restricted_step.py:64-120, in ONE
 * call: the 1-D search over the step-length parameter alpha of the family `st` until the constraint measure of the
 * total step s(alpha) + scons equals the radius delta — same start value, bracket updates, Newton / bisection
 * schedule (bisection only once niter > 4 unless newton_safe), nextafter bracket test and tolerances as the
 * reference, so the sequence of trial alphas is the reference's.
 *   cons: 0 trust region |s| (TrustRegion.cons :136-142); 1 largest per-atom displacement (RestrictedAtomicStep.cons
 *         :172-183, nout = 3 natoms); 2 largest weighted component |w_i s_i| (MaxInternalStep.cons :206-216);
 *         3 weighted sphere |(s + d1) * w| (IRCTrustRegion.cons :152-158).
 *   scons (nout) or NULL: the constraint-correction step added to every trial step (:35-37, :73-76);
 *   w (nout) for cons 2 / 3, d1 (nout) for cons 3;
 *   alpha0, alphamin, alphamax, slope, newton_safe: the family's class attributes (stepper.py:20-41);
 *   orthonormal != 0: the columns of the stepper's eigenvector matrix are orthonormal in the OUTPUT space, so for
 *         cons 0 the norm is evaluated in the eigenbasis (|s|^2 = |shat|^2 + 2 shat.V^T scons + |scons|^2) and no
 *         device work at all happens per trial alpha.
 *   sel (m ints) / nfull: the family lives in the subspace of the free coordinates sel[0..m) of an nfull-dimensional
 *         space (projection basis = columns of the identity, constraints that pin single coordinates); scons, w, d1,
 *         the measure and the returned step are then nfull-dimensional.  NULL / 0 otherwise.
 * Per trial alpha otherwise: O(m) host arithmetic, one 2-right-hand-side device matvec and one single-workgroup
 * reduction; only two scalars come back.  Once the schedule is pure bisection (sixth trial on for the families that are
 * not newton_safe, restricted_step.py:100-110) the next four levels of midpoints — 15 trial alphas, the reference's own —
 * are evaluated per round trip (option rs_batch): secular roots per candidate on the device, ONE pass over the
 * eigenvector matrix for all of them on the matrix cores, one measure per workgroup.  Outputs: s (nout) = total step at the final alpha, *val = the measure
 * reported by the reference (the value itself inside the radius, delta on the boundary), alphas[0 .. *nalpha) the
 * trial sequence (capacity maxiter + 1; may be NULL).  Returns SELLA_E_NOCONV where the reference raises
 * RuntimeError("Restricted step failed to converge!").                                                          */
int sella_restricted_step(sella_stepper* st, int cons, double delta, const double* scons, const double* w,
                          const double* d1, double alpha0, double alphamin, double alphamax, double slope,
                          int newton_safe, int orthonormal, double tol, int maxiter, const int* sel, int nfull,
                          double* s, double* val, double* alphas, int* nalpha);

/* ---- one optimizer step per call ------------------------------------------------------------- */
/* Sella.step (sella/optimize/optimize.py:359-440) between two force calls, in ONE call: PES.kick's model prediction,
 * ratio and quasi-Newton update (peswrapper.py:578-602, linalg.py:274-304), the trust-radius rule (optimize.py:413-434)
 * and the restricted step at the new point (restricted_step.py:28-120, stepper.py) — for the Cartesian PES whose
 * approximate Hessian carries a STRUCTURED eigendecomposition (sella_update_h_lr), unconstrained or with constraints
 * that pin single coordinates (idx / m: the free coordinates; the view Bsub = B[idx][idx] with its own structured
 * decomposition as in sella_update_h_lr), constraints satisfied (no correction step).  The calculator boundary
 * (peswrapper.py:413-418) stays with the caller: it moves the atoms by the step returned, evaluates f and g, calls again.
 * Every phase runs the routines of the one-phase entry points, so results are theirs bit for bit.
 *   flags: SELLA_OPT_LEARN  uses dx (step taken, n), g_old, g_new, f_old, f_new, smag (measure of the step taken as
 *                           reported by the restricted step), delta / rho (in-out), the five radius parameters;
 *                           outputs df_pred, ratio + ratio_valid (|df_pred| >= 1e-14), nrank1, nrank1_sub, *r, mu, ...
 *          SELLA_OPT_PROPOSE uses g_new, delta, stepper_kind (SELLA_STEP_*), order, cons (0 tr, 1 ras), tol, maxiter;
 *                           outputs s_out (n), smag_out, nalpha.
 * The re-diagonalisation schedule (optimize.py:363-378) stays with the caller: it asks for LEARN only, runs the
 * iterative diagonalisation, then asks for PROPOSE.                                                                  */
enum { SELLA_OPT_LEARN = 1, SELLA_OPT_PROPOSE = 2 };
typedef struct sella_opt_step_t {
    int flags, n;
    /* approximate Hessian and its structured eigendecomposition (sella_update_h_lr) */
    sella_mat B, Wt;
    int* r;
    double* mu;
    double lam0;
    int update_method, symm;
    int B_stale, Bsub_stale;   /* in-out: the dense mirrors lag behind (W, mu, lam0); rebuilt by sella_lr_materialize when needed */
    /* principal-submatrix view of pinned-coordinate constraints (idx == NULL: none) */
    sella_mat Bsub, Wt_sub;
    int* r_sub;
    double* mu_sub;
    const int* idx;
    int m;
    /* learn */
    const double *dx, *g_old, *g_new;
    double f_old, f_new, smag;
    double delta, rho;                                      /* in-out */
    double delta_min, sigma_inc, sigma_dec, rho_inc, rho_dec;
    double df_pred, ratio;                                  /* out */
    int ratio_valid, updated, nrank1, nrank1_sub;           /* out; updated = 0: step shorter than 1e-8, B left alone */
    /* propose */
    int stepper_kind, order, cons, maxiter;
    double tol;
    double* s_out;                                          /* n */
    double smag_out;                                        /* out */
    int nalpha;                                             /* out */
    double alpha_hint;         /* in-out: alpha at which the previous root search of THIS search ended (0: none): the first
                                * round of the batched search looks around it; same root, fewer rounds                    */
} sella_opt_step_t;
int sella_opt_step(sella_ctx* ctx, sella_opt_step_t* io);
/* Dense mirror of a structured decomposition: B <- lam0 I + W^T diag(mu - lam0) W (B n x n resident, overwritten).  The
 * fast form of sella_opt_step updates (W, mu) only and marks B stale; whoever needs the matrix itself — `H.B`, a matvec, a
 * projection, the general update route — rebuilds it first.                                                             */
int sella_lr_materialize(sella_ctx* ctx, sella_mat B, sella_mat Wt, int r, const double* mu, double lam0);

/* ---- internal-coordinate primitives ----------------------------------------------------------- */
/* Batched value / gradient / Hessian-vector product / Hessian of bonds (natoms = 2), angles (3) and
 * dihedrals (4): the vmapped JAX functions of sella/internal.py:58-135
 *   _bond_value/_angle_value/_dihedral_value (:58-80), *_grad_batched (:85-87),
 *   *_hess_batched (:95-97), *_hvp_batched (:106-135).
 * pos (nc, natoms, 3) gathered atom positions; tvec (nc, natoms-1, 3) periodic shift vectors or NULL;
 * tangent (nc, natoms, 3) or NULL.  Outputs (host): q (nc), grad (nc, natoms, 3),
 * hvp (nc, natoms, 3) when tangent is given, hess (nc, 3 natoms, 3 natoms) when not NULL.            */
int sella_internals_eval(sella_ctx* ctx, int natoms, int nc, const double* pos, const double* tvec,
                         const double* tangent, double* q, double* grad, double* hvp, double* hess);

/* ---- EMT calculator (far side of the calculator boundary, sella/peswrapper.py:413-418) -------------- */
/* Energy and gradient of the effective-medium potential (functional form of ase/calculators/emt.py).
 * pos (n x 3); par (9 x n): per-atom E0, s0, V0, eta2, kappa, lambda, n0, gamma1, gamma2 in eV / Angstrom;
 * shifts (nshift x 3): lattice translations of the periodic images to include (with the zero vector);
 * rc, acut, cutoff, beta: cutoff function parameters.  Outputs: *energy, grad (n x 3) = dE/dx.
 * Limits (also of sella_calc_emt_create): n < 2^24 atoms and nshift <= 127 images (SELLA_E_INVALID beyond: the density pass
 * hands each thread's neighbours to the force pass as packed (atom, image) words); the hand-over area is 256 x 9 ints =
 * 9.2 KB of scratch per atom, written and read by every force call (9.4 MB at 1024 atoms).                            */
int sella_emt_eval(sella_ctx* ctx, int n, const double* pos, const double* par, int nshift,
                   const double* shifts, double rc, double acut, double cutoff, double beta,
                   double* energy, double* grad);

/* ---- calculators that live in the library, and the finite-difference Hessian on top of one ----------------- */
/* sella/peswrapper.py:413-418 evaluates energy and forces through `atoms.calc`; for a calculator implemented HERE that
 * boundary can be crossed without the host language: sella_calc_eval(calc, x, &f, g) -> energy and gradient dE/dx.
 *   model: f(x) = 1/2 x^T A x + c/3 sum_j (u_j . x)^3, A (n x n resident), U (nu x n host, copied);
 *   emt:   the arguments of sella_emt_eval, fixed at creation (n = 3 natoms).                                        */
typedef struct sella_calc sella_calc;
int sella_calc_model_create(sella_ctx* ctx, sella_mat A, const double* U, int nu, int n, double c, sella_calc** calc);
int sella_calc_emt_create(sella_ctx* ctx, int natoms, const double* par, int nshift, const double* shifts, double rc,
                          double acut, double cutoff, double beta, sella_calc** calc);
int sella_calc_eval(sella_calc* calc, const double* x, double* energy, double* grad);
long sella_calc_ncalls(sella_calc* calc);
int sella_calc_dim(sella_calc* calc);
int sella_calc_destroy(sella_calc* calc);
/* NumericalHessian (sella/linalg.py:14-101): H v ~ scale (g(x0 + eta v / scale) - g0) / eta (threepoint: the central
 * form), scale = +-|v| by the reference's orientation rule (:45-73), seen through the free coordinates idx[0..m) of a
 * pinned-coordinate constraint set (NULL: all n).  sella_fd_matvec has the sella_matvec_fn signature with the operator
 * as `user`, so sella_davidson(ctx, SELLA_NO_MAT, sella_fd_matvec, fd, ...) is the iterative diagonalisation of
 * peswrapper.py:508-556 with no host-language frame between its force calls.  Every product is remembered:
 * sella_fd_pairs returns the displaced directions and difference quotients, (n x k) row-major each, k = sella_fd_npairs,
 * which PES.diag turns into secant pairs for the Hessian update (peswrapper.py:545-553).                              */
typedef struct sella_fd sella_fd;
int sella_fd_create(sella_calc* calc, int n, const double* x0, const double* g0, double eta, int threepoint,
                    const int* idx, int m, sella_fd** fd);
int sella_fd_matvec(void* fd, const double* v, double* Av, int m);
int sella_fd_npairs(sella_fd* fd);
long sella_fd_calls(sella_fd* fd);
int sella_fd_pairs(sella_fd* fd, double* Vs, double* AVs);
int sella_fd_destroy(sella_fd* fd);

/* ---- a whole search in the library --------------------------------------------------------------------------- */
/* `Sella(atoms, ...).run(fmax, steps)` (sella/optimize/optimize.py:42-440 under ASE's Optimizer.irun) for the
 * configuration an ensemble of independent searches consists of (BASELINE configs[3]): Cartesian PES, no constraints or
 * pinned coordinates (idx / m: the free ones), a calculator of the library (sella_calc_*), TS-BFGS, structured
 * approximate Hessian, built-in step family and measure, eig = True.  The loop is the reference's — first-use
 * diagonalisation (:318-326), restricted step, kick, re-diagonalisation rule (:363-378), trust-radius rule (:413-434),
 * convergence on the largest per-atom projected force (peswrapper.py:438-441) — made of the entry points above
 * (sella_davidson over sella_fd_matvec, sella_update_h_lr, sella_lr_restrict, sella_opt_step); the host language is not
 * entered between creation and return, so one host thread per replica can drive one GPU.  SELLA_E_UNSUPPORTED: the
 * search left the covered configuration (explicit rank beyond 0.4 n, ...): the caller continues with the general driver.
 *   cons: 0 trust region (delta0 = per-coordinate radius x free coordinates, optimize.py:183-186), 1 per-atom measure.  */
typedef struct sella_search sella_search;
typedef struct sella_search_params_t {
    int order, eig, threepoint, dav_method, stepper_kind, cons, update_method, symm, nsteps_per_diag;
    long diag_every_n;                         /* < 0: never */
    double eta, gamma, delta0, delta_min, sigma_inc, sigma_dec, rho_inc, rho_dec;
} sella_search_params_t;
int sella_search_create(sella_ctx* ctx, sella_calc* calc, int n, const double* x0, const int* idx, int m,
                        const sella_search_params_t* params, sella_search** search);
/* energy and gradient (dE/dx, n) at the starting point, if the caller evaluated them already (counts as a force call) */
int sella_search_seed(sella_search* search, double energy, const double* grad);
int sella_search_run(sella_search* search, double fmax, long steps, int* converged);
/* After SELLA_E_UNSUPPORTED: the block of secant pairs of a diagonalisation that no longer fitted the structured form (its
 * force calls are spent and counted; when an optimizer step scheduled the diagonalisation, that step has moved the geometry
 * and is counted): *k pairs (0: none pending), Sm and Ym (n x k row-major; NULL: only the count).  The caller applies them
 * as one block update of the approximate Hessian it takes over (sella/linalg.py:274-304) and is then exactly where the
 * reference is after PES.diag (sella/peswrapper.py:545-553).  Two hand-overs leave the geometry where it was and count
 * no step: k = 0 (the capacity check in front of a step), and the FIRST-USE diagonalisation of a search (k > 0 with
 * counters[0] == 0: optimize.py:318-326 runs it before the first step is taken).                                       */
int sella_search_pending_pairs(sella_search* search, int* k, double* Sm, double* Ym);
/* x, g (n each, may be NULL); scalars[5] = f, fmax, delta, rho, lowest eigenvalue of the approximate Hessian;
 * counters[6] = optimizer steps, force calls, one-call steps, explicit rank, explicit rank of the view (-1: none),
 * first-use diagonalisation done (0 / 1)                                                                              */
int sella_search_state(sella_search* search, double* x, double* g, double* scalars, long* counters);
/* Hand the approximate Hessian over to the caller, who continues with the general driver: the matrix handles change owner
 * (the search cannot be run again).  mats[4] = B, Wt, Bsub, Wt_sub (SELLA_NO_MAT where absent); ints[8] = r (-1: no
 * Hessian yet), r_sub (-1: no view), rows of Wt, rows of Wt_sub, B_stale, Bsub_stale (the dense matrices lag behind the
 * decompositions: sella_lr_materialize), steps since the last diagonalisation, first_diag; mu / mu_sub: `rows` entries.  */
int sella_search_release_hessian(sella_search* search, sella_mat* mats, long* ints, double* mu, double* mu_sub,
                                 double* lam0);
int sella_search_destroy(sella_search* search);
sella_ctx* sella_search_ctx(sella_search* search);        /* the context the search lives on */

/* ---- cohorts: the replica dimension of the ensemble (BASELINE configs[3]) ------------------------- */
/* The reference's ensemble members are independent `Sella` objects (sella/optimize/optimize.py:42-81, 359-440).  A cohort
 * advances up to 16 of them on ONE GPU in lockstep from ONE host thread: every member keeps its own context (memory,
 * rings, scratch), but while the cohort runs the members share one stream, every kernel of the step is launched ONCE for
 * all members that have reached it (the member is the grid's z index, its arguments a by-value descriptor array), and
 * every wait is one stream synchronisation for all of them.  Members whose control flow differs (a further round of the
 * root search, a re-diagonalisation) form launches of their own and rejoin at the next step boundary.  A member's results
 * are bit-identical to sella_search_run on its own.
 *   sella_cohort_create: `members` = n distinct contexts of one device (1 <= n <= 16), each in no other cohort; they stay
 *     usable on their own between runs.  Destroy the cohort before its contexts.
 *   sella_cohort_run_searches: searches[i] lives on member context i (NULL: empty slot); converged[i] / status[i] are what
 *     sella_search_run(searches[i], fmax, steps, ...) would have returned (SELLA_E_UNSUPPORTED: that member left the
 *     covered configuration, see sella_search_pending_pairs); sella_cohort_error(cohort, i) is its message.
 *   sella_cohort_stats: counters[8] = scheduler rounds, launches asked for by the members, launches issued (merged),
 *     waits asked for, stream synchronisations, barrier arrivals, microseconds spent inside the members' host code,
 *     microseconds spent issuing the merged launches — accumulated since creation.                                     */
typedef struct sella_cohort sella_cohort;
int sella_cohort_create(sella_ctx* const* members, int n, sella_cohort** cohort);
int sella_cohort_size(sella_cohort* cohort);
int sella_cohort_run_searches(sella_cohort* cohort, sella_search* const* searches, int n, double fmax, long steps,
                              int* converged, int* status);
/* on != 0: the members' host code between their launches runs on a worker thread each instead of on fibers of the calling
 * thread (parallel host code; the launches are still merged and issued by the caller of sella_cohort_run_searches); width + 1
 * cores spin per cohort while it runs.  Results do not depend on the mode.                                            */
int sella_cohort_member_threads(sella_cohort* cohort, int on);
int sella_cohort_stats(sella_cohort* cohort, long* counters);
const char* sella_cohort_error(sella_cohort* cohort, int member);
int sella_cohort_destroy(sella_cohort* cohort);

/* ---- profiling hooks (bench.py roofline leg) ---------------------------------------------- */
/* When enabled, every launch of the big streaming kernels is bracketed by hipEvents on the
 * context stream.  kind: 0 = row-panel matvec (n x n streams), 1 = gemm, 2 = update, 3 = other,
 * 4 = small matvecs (panel dots), 5 = trailing-matrix matvec of the tridiagonalisation (eigh).     */
int sella_prof_enable(sella_ctx* ctx, int on);
int sella_prof_reset(sella_ctx* ctx);
int sella_prof_get(sella_ctx* ctx, int kind, long* launches, double* total_ms,
                   double* total_bytes, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif /* SELLA_HIP_H */
