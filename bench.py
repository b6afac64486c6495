#!/usr/bin/env python3
"""Benchmark of the Sella inner saddle-point linear-algebra loop on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run; the ranks
                                                             talk through librccl bound with ctypes, sella_amd/comm.py)

Metric (BASELINE.json): Davidson iterations per second on the synthetic 3N = 3072 Hessian of
SURVEY.md §8(d).  One *step* = one complete `rayleigh_ritz(A, gamma=0.1, P, v0=g, 'jd0',
maxiter=40)` call of the HIP path with A and P already resident in HBM — including the device
eigendecomposition of the preconditioner P that the eigenbasis form of the JD correction needs
once per call (the reference pays a dense LU per iteration instead).  `value` = Davidson
iterations (vectors added to the subspace, the unit BASELINE.md quotes) summed over all ranks,
divided by the slowest rank's time for the K steps.  The steps run through `--seeds` (4) x `--starts` (5) independent
problems of the same recipe, because the exit iteration of one gamma = 0.1 run is chaotic (20 ... 31
vectors for the same matrices) and the per-call eigh is amortised over it.  Ranks are independent replicas (the path has
no exchange step), so scaling is weak.

Extra objects on the JSON line:
  roofline      the dominant kernel of the step (`trd_gemv_kernel`, the trailing-matrix matvec of the eigh of P):
                algorithmic bytes per launch (8 m^2) / mean launch time from hipEvents attached to the dispatch
                packets on the library's own stream, collected in a separate instrumented pass; the Davidson
                loop's own n x n streams are reported beside it.
  block_davidson  BASELINE configs[4]: block Davidson (16 vectors per iteration, H.V panel on the matrix cores) at
                3N = 12288, rows of H sharded over the ranks (strong scaling), time per block iteration.
  collective    which binding carried the multi-rank exchanges (direct RCCL or the torch.distributed fallback).
  cpu_baseline  the NumPy/SciPy oracle port of the reference algorithm (dense LU per iteration)
                timed on this box's host cores on the same inputs (one call), rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)


def hessian_like(n, seed, eps=5e-3):
    """one negative mode, log-uniform positive spectrum, noisy preconditioner."""
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.exp(rng.uniform(np.log(0.05), np.log(50.0), n))
    lam[0] = -1.0
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    N = rng.normal(size=(n, n))
    P = A + eps * 0.5 * (N + N.T)
    g = rng.normal(size=n)
    return A, P, g


class EnsembleMember:
    """Picklable factory of the ensemble leg's members (BASELINE configs[3]): `prepare(i)` builds the host-side data
    (outside the timed region), `factory(i)` uploads the model Hessian to the calling process's device context and
    returns the Atoms object."""

    def __init__(self, ne):
        self.ne = ne
        self.host = {}

    def prepare(self, i):
        rngi = np.random.RandomState(6000 + i)
        Ui = rngi.normal(size=(8, self.ne))
        Ui /= np.linalg.norm(Ui, axis=1)[:, None]
        self.host[i] = (hessian_like(self.ne, seed=5000 + i)[0], Ui, 0.05 * rngi.normal(size=(self.ne // 3, 3)))

    SELLA_KW = dict(order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', proj_trans=False)

    def warmup(self):
        """Called once per worker process before the clock: a short search on a member outside the ensemble's index
        range, so that code objects, scratch and pinned buffers exist — what the earlier legs do for the parent."""
        from sella_amd.ensemble import run_one
        run_one(self(-1), 0.0, 3, self.SELLA_KW)
        self.host.pop(-1, None)          # (several threads may warm up at once)

    def __call__(self, i):
        from sella_amd import device as _dev
        from sella_amd.atoms import Atoms, QuadraticCubicModel
        if i not in self.host:
            self.prepare(i)
        Ai, Ui, x0 = self.host[i]
        c_i = _dev.get_context()
        dAi = c_i.upload(Ai)
        at = Atoms(['X'] * (self.ne // 3), x0.copy(), pbc=True)
        at.calc = QuadraticCubicModel(lambda x, c_i=c_i, dAi=dAi: c_i.symm_mm(dAi, x), Ui, c=0.05, device_matrix=dAi)
        return at


class EmtSlabMember:
    """configs[3] as BASELINE names it: member i = 256-atom Cu(111) EMT slab (8 x 8 x 4) with a lifted surface atom, lower
    half pinned atom by atom, thermal jitter from seed i (3N = 768, 384 free coordinates, device EMT, default `Sella`
    keywords); returns (atoms, the member's own keywords)."""
    SELLA_KW = dict(order=1, eta=1e-4, gamma=0.1)

    def __init__(self):
        self.host = {}

    def prepare(self, i):
        """Host-side description of member i (geometry, constraint list, calculator object: interpreter work, serial
        under the interpreter lock) — outside the timed region, like `EnsembleMember.prepare`; everything that touches
        the device (first force call, library calculator, the search) happens inside it."""
        self.host[i] = self._build(i)

    def __call__(self, i):
        built = self.host.pop(i, None)           # a prepared member is consumed: the next pass builds its own
        return built if built is not None else self._build(i)

    def _build(self, i):
        from sella_amd import Constraints
        from sella_amd.atoms import EMT, fcc111
        slab = fcc111('Cu', (8, 8, 4), vacuum=7.5)
        slab.positions += 0.02 * np.random.RandomState(100 + i).normal(size=slab.positions.shape)
        top = int(np.argmax(slab.positions[:, 2]))
        site = slab.info['adsorbate_sites']['bridge']
        slab.positions[top] += np.array([site[0], site[1], 1.9])
        cons = Constraints(slab)
        for a in slab:
            if a.position[2] < slab.cell[2, 2] / 2.:
                cons.fix_translation(a.index)
        slab.calc = EMT()
        return slab, dict(constraints=cons)

    def warmup(self):
        from sella_amd.ensemble import run_one
        run_one(self(-1), 0.0, 3, self.SELLA_KW)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--n', type=int, default=3072)
    ap.add_argument('--maxiter', type=int, default=40)
    ap.add_argument('--gamma', type=float, default=0.1)
    ap.add_argument('--seeds', type=int, default=4)
    ap.add_argument('--starts', type=int, default=5)
    ap.add_argument('--converged-n', type=int, default=768)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--option', action='append', default=[], metavar='KEY=VALUE',
                    help='library tuning knob for this run (sella_ctx_set_option; recorded on the line), repeatable')
    ap.add_argument('--no-optimizer', action='store_true')
    ap.add_argument('--opt-steps', type=int, default=20)
    ap.add_argument('--emt-steps', type=int, default=10)
    ap.add_argument('--ensemble-per-gpu', type=int, default=8)
    ap.add_argument('--ensemble-steps', type=int, default=20)
    ap.add_argument('--ensemble-n', type=int, default=768)
    ap.add_argument('--ensemble-threads', type=int, default=-1,
                    help='host threads per GPU for the ensemble leg (persistent contexts, sella_amd.ensemble.EnsembleThreads); '
                         '-1: min(members per GPU, 8, 2 x CPUs per rank); 0 / 1: none')
    ap.add_argument('--ensemble-reps', type=int, default=3,
                    help='complete passes of the ensemble leg; the median one is reported, all are listed')
    ap.add_argument('--ensemble-emt', type=int, default=8,
                    help='members of the second ensemble figure: 256-atom EMT slab searches, configs[3] as named (0: skip)')
    ap.add_argument('--ensemble-saturate', type=int, default=4,
                    help='second ensemble figure with this many times the members per GPU on up to 12 threads (0: skip)')
    ap.add_argument('--ensemble-procs', type=int, default=-1,
                    help='worker processes per GPU for the ensemble leg (-1: min(members, 4, CPUs of this rank); 0/1: none)')
    ap.add_argument('--concurrent', type=int, default=4,
                    help='extra leg: this many independent problems in flight on host threads (0 / 1: skip)')
    ap.add_argument('--block-n', type=int, default=12288, help='configs[4] leg: operator size (0 disables)')
    ap.add_argument('--block-iters', type=int, default=12)
    ap.add_argument('--block-eigh', type=int, default=1, help='time the full device eigh of the block operator at N = 1 (0: skip)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

    from sella_amd import comm as comm_mod
    from sella_amd.device import Context
    from sella_amd.utilities.hostcpu import effective_cpu_count, limit_blas_threads
    # host threads: the CPUs this container may use (cgroup quota, not the visible count), shared by the ranks
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    host_threads = limit_blas_threads(max(1, effective_cpu_count() // max(1, local_world)))
    ctx = Context()             # LOCAL_RANK selects the device (one process per GPU)
    for kv in args.option:
        key, _, val = kv.partition('=')
        ctx.set_option(key, int(val))
    # Collectives: librccl bound with ctypes on this context's stream and buffers (sella_amd/comm.py) — no PyTorch.
    # The CPU tests (gloo, host emulation) set SELLA_BENCH_COMM=gloo; if the direct binding cannot come up on a GPU
    # box, torch.distributed's nccl backend (the same RCCL) is the fallback, and the line says which was used.
    comm = comm_mod.SingleProcess()
    if world > 1:
        def torch_group(backend):
            import torch
            import torch.distributed as tdist
            if backend == 'nccl':
                torch.cuda.set_device(local_rank)
            tdist.init_process_group(backend=backend, rank=rank, world_size=world)
            return comm_mod.GlooCommunicator()
        if os.environ.get('SELLA_BENCH_COMM') == 'gloo':
            comm = torch_group('gloo')
        else:
            try:
                comm = comm_mod.RcclCommunicator(ctx, rank, world)
            except Exception as e:                       # noqa: BLE001 — any failure of the direct binding
                sys.stderr.write(f'[bench rank {rank}] direct RCCL binding failed ({e}); using torch.distributed\n')
                comm = torch_group('nccl')
        comm_mod._comm = comm                            # the ensemble / sharded-operator code uses the same one
    n = args.n
    # The exit iteration of the gamma = 0.1 run is chaotic (20 ... 31 vectors for the
    # same matrix depending on the last bit of the arithmetic), and the fixed eigh cost is amortised over
    # it, so the steps run through `args.seeds` independent matrices of the same recipe, each with
    # `args.starts` start vectors (the recipe's own gradient first, then further draws of the same
    # distribution) — 20 distinct problems by default, one per step — and the reported rate is the average
    # over them: a one-ulp change anywhere in the arithmetic moves single exits by +-5 iterations, the mean
    # over 20 by about one.  Seed 0 / start 0 of rank 0 is the problem the CPU baseline runs.
    mats = []
    for sd in range(args.seeds):
        A_s, P_s, g_s = hessian_like(n, seed=rank * args.seeds + sd)
        mats.append((ctx.upload(A_s), ctx.upload(P_s), g_s))
        if sd == 0:
            A, P, g = A_s, P_s, g_s
    problems = []
    for st in range(args.starts):
        for sd in range(args.seeds):
            g_v = mats[sd][2] if st == 0 else np.random.RandomState(
                777000 + 1000 * (rank * args.seeds + sd) + st).normal(size=n)
            problems.append((mats[sd][0], mats[sd][1], g_v))
    dA, dP = problems[0][0], problems[0][1]
    step_no = [0]

    def one_step():
        dA_s, dP_s, g_s = problems[step_no[0] % len(problems)]
        step_no[0] += 1
        w, V, Vt = ctx.eigh(dP_s)
        lams, Vr, AVr, nmv = ctx.davidson(dA_s, n, g_s, args.gamma, method='jd0', maxiter=args.maxiter,
                                          Pvecs=V, PvecsT=Vt, pevals=w)
        V.free()
        Vt.free()
        return lams, Vr, AVr, nmv

    def barrier():
        ctx.sync()
        comm.barrier()

    def settle():
        # Behind a leg that ran several host threads with a device context each, the calls of this process are slower for
        # a few hundred milliseconds (tools/aftermath_probe.py: a 15-vector Davidson call 1.16 ms fresh, 1.34 ms right
        # after four such threads have released ~4 GB of device memory, 1.15 ms one second later): the single-threaded legs
        # that follow wait for that to pass.
        ctx.sync()
        time.sleep(1.0)

    for _ in range(args.warmup):
        one_step()
    barrier()
    step_no[0] = 0
    t0 = time.perf_counter()
    iters = 0
    first = None
    for _ in range(args.steps):
        out = one_step()
        iters += out[1].shape[1]
        if first is None:
            first = out                             # problem 0: the one the parity block refers to
    lams, Vr, AVr, nmv = first
    ctx.sync()
    elapsed = time.perf_counter() - t0
    barrier()

    # Davidson loop alone (P's eigendecomposition kept from a previous optimizer phase).  Measured here, in the state the
    # timed loop above ran in: behind the threaded leg below every call of this process is ~0.16 ms slower for a while
    # (11.2 k against 12.7 k iterations/s, round 6: settle()).
    w, V, Vt = ctx.eigh(dP)
    settle()
    t1 = time.perf_counter()
    it2 = 0
    for _ in range(args.steps):
        _, Vr2, _, _ = ctx.davidson(dA, n, g, args.gamma, method='jd0', maxiter=args.maxiter,
                                    Pvecs=V, PvecsT=Vt, pevals=w)
        it2 += Vr2.shape[1]
    ctx.sync()
    t_loop = time.perf_counter() - t1
    # ---- the same steps, several independent problems in flight (host threads, a device context and stream each) -------
    # One rayleigh_ritz call is a chain of ~6,000 dependent launches of a few workgroups: it leaves most of the chip (and
    # the host's other cores) idle.  Independent saddle searches — the ensemble of configs[3], or the 20 problems of this
    # leg — fill it.  Reported BESIDE the headline (`value` stays the one-problem-at-a-time rate).
    concurrent = None
    if args.concurrent > 1 and world == 1:
        import threading
        from sella_amd import device as _devm
        T = args.concurrent
        per = max(1, args.steps // T)
        done, errs = [0] * T, []
        host = [hessian_like(n, seed=t_ % args.seeds) for t_ in range(T)]

        def work(t_):
            cx = dA_t = dP_t = None
            try:
                cx = Context()                       # (created by the thread that uses it: it owns the handles it frees)
                _devm.use_context(cx)
                A_s, P_s, g_t = host[t_]
                dA_t, dP_t = cx.upload(A_s), cx.upload(P_s)
                for i_ in range(per + 1):
                    if i_ == 1:
                        gate.wait()                  # (first pass = warm-up; the clock starts when every thread is warm)
                    w_t, V_t, Vt_t = cx.eigh(dP_t)
                    r_ = cx.davidson(dA_t, n, g_t if i_ % 2 == 0 else g_t[::-1].copy(), args.gamma, method='jd0',
                                     maxiter=args.maxiter, Pvecs=V_t, PvecsT=Vt_t, pevals=w_t)
                    V_t.free()
                    Vt_t.free()
                    if i_ >= 1:
                        done[t_] += r_[1].shape[1]
                cx.sync()
            except BaseException as e:               # noqa: BLE001 — reported on the line
                errs.append(repr(e)[:200])
                try:
                    gate.abort()
                except Exception:                    # noqa: BLE001
                    pass
            finally:
                _devm.use_context(None)
                if cx is not None:
                    import gc
                    del dA_t, dP_t
                    gc.collect()
                    cx.close()
        gate = threading.Barrier(T + 1)
        ths = [threading.Thread(target=work, args=(t_,)) for t_ in range(T)]
        for th in ths:
            th.start()
        try:
            gate.wait()
            tc0 = time.perf_counter()
        except threading.BrokenBarrierError:
            tc0 = time.perf_counter()
        for th in ths:
            th.join()
        tcc = time.perf_counter() - tc0
        concurrent = dict(problems_in_flight=T, calls=per * T, davidson_iter_per_s=round(sum(done) / tcc, 1),
                          ms_per_call_amortised=round(1e3 * tcc / (per * T), 3), errors=errs or None)
        settle()

    # What the drop-in call `rayleigh_ritz(A_numpy, gamma, P_numpy)` pays on top of `value`: A and P go up over PCIe
    # (2 x 8 n^2 bytes, pageable numpy memory as the reference's callers hold it) — measured, not priced from the spec
    dtmp = [ctx.upload(A), ctx.upload(P)]
    ctx.sync()
    for m_ in dtmp:
        m_.free()
    tu0 = time.perf_counter()
    dtmp = [ctx.upload(A), ctx.upload(P)]
    ctx.sync()
    upload_ms = 1e3 * (time.perf_counter() - tu0)
    for m_ in dtmp:
        m_.free()
    t2 = time.perf_counter()
    for _ in range(3):
        w_, V_, Vt_ = ctx.eigh(dP)
        V_.free()
        Vt_.free()
    ctx.sync()
    t_eigh = (time.perf_counter() - t2) / 3
    # parity evidence carried on the line: lowest Ritz pair of the benchmark call against the host, and
    # the north_star criterion proper — the CONVERGED lowest eigenvalue (gamma = 1e-7, run to convergence)
    # against the exact one, which is -1 by construction of the synthetic Hessian
    resid = float(np.linalg.norm(A @ Vr[:, 0] - lams[0] * Vr[:, 0]))
    av_err = float(np.abs(AVr - A @ Vr).max())
    conv = None
    if rank == 0 and world == 1 and args.converged_n > 0:
        # (done at 3N = 768, the ensemble-member size: the Rayleigh-Ritz algebra on the host is O(k^3) per
        # iteration in scalar C++, so the ~500-vector run at 3N = 3072 takes 90 s; it is a -m gpu test,
        # tests/test_big_gpu.py::test_davidson_converged_eigenpair, 3e-15 there)
        nc = args.converged_n
        Ac, Pc, gc = hessian_like(nc, seed=0)
        dAc, dPc = ctx.upload(Ac), ctx.upload(Pc)
        wc_, Vc0, Vtc0 = ctx.eigh(dPc)
        tcv = time.perf_counter()
        lc_, Vc_, _, _ = ctx.davidson(dAc, nc, gc, 1e-7, method='jd0', maxiter=900, Pvecs=Vc0, PvecsT=Vtc0, pevals=wc_)
        conv = dict(n=nc, gamma=1e-7, vectors=int(Vc_.shape[1]), seconds=round(time.perf_counter() - tcv, 3),
                    lowest_eigenvalue_abs_err=float(abs(lc_[0] + 1.0)),
                    residual_norm=float(np.linalg.norm(Ac @ Vc_[:, 0] - lc_[0] * Vc_[:, 0])))
        for m_ in (dAc, dPc, Vc0, Vtc0):
            m_.free()

    # ---- roofline: instrumented pass over whole steps (hipEvents on the library stream) ------------
    # By time the dominant kernel of a step is the trailing-matrix matvec of the tridiagonalisation
    # (`trd_gemv_kernel`, one launch per column of P, HBM bound: 8*m^2 algorithmic bytes for the
    # m x m trailing block); the Davidson loop's own n x n streams are reported beside it.
    ctx.prof_reset()
    ctx.prof_enable(True)
    nprof = max(1, min(args.steps, 3))
    for _ in range(nprof):
        one_step()
    ctx.prof_enable(False)
    pt = ctx.prof_get(5)          # trd_gemv_kernel
    pg = ctx.prof_get(0)          # n x n streams: gemv_rows_kernel<1,2> (A t) and <2,2> (Q^T[r v], Q[a b])
    ps = ctx.prof_get(4)          # panel dots: same template, <*,1> instantiations, k x n, latency bound

    def hbm(p, kernel):
        if p['launches'] <= 0 or p['ms'] <= 0:
            return None
        achieved = p['bytes'] / (p['ms'] * 1e-3) / 1e9
        return dict(bound='hbm', kernel=kernel, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None, launches=p['launches'],
                    mean_us=round(1e3 * p['ms'] / p['launches'], 2),
                    bytes_per_launch=round(p['bytes'] / p['launches']))

    roof = hbm(pt, 'trd_gemv_kernel (m x m trailing-matrix matvec, one per column of the eigh of P)')
    if roof is None:
        # a matrix too small for the blocked chain of the eigensolver (at most 1024 trailing rows go through the
        # one-launch-per-column chain, csrc/eigh.hip): the Davidson loop's own n x n stream is the kernel to report
        roof = hbm(pg, 'gemv_rows_kernel<NRHS,2> (n x n row-panel matvec of the Davidson loop; n below the blocked chain of eigh)')
    elif roof is not None:
        # HBM traffic per launch cannot be counted from inside this process: it comes from the PMC
        # passes of tools/gpu_session.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, own runs), whose
        # per-launch means are committed in profiles/pmc_traffic.json with the guide's gfx950 correction
        try:
            with open(os.path.join(REPO, 'profiles', 'pmc_traffic.json')) as f:
                pmc = json.load(f)['trd_gemv_kernel']
            if pmc['n'] == n:
                roof['traffic'] = round(pmc['bytes_per_launch'])
                roof['traffic_source'] = pmc['source']
        except (OSError, KeyError, ValueError):
            pass
        # every 4th column of the tridiagonalisation is instrumented (csrc/eigh.hip: the queue is drained before an
        # instrumented column so that the packet time stamps are the kernel's own): the sampled time is a quarter of
        # the kernel's share of a step
        roof['sampled_every'] = 4
        roof['sampled_ms_per_step'] = round(pt['ms'] / nprof, 2)
        roof['share_of_step_ms'] = round(4.0 * pt['ms'] / nprof, 2)
        roof['davidson_matvec'] = hbm(pg, 'gemv_rows_kernel<NRHS,2> (n x n row-panel matvec of the Davidson loop)')
        roof['davidson_panel_dots'] = dict(launches=ps['launches'],
                                           mean_us=round(1e3 * ps['ms'] / max(1, ps['launches']), 2),
                                           bytes_per_launch=round(ps['bytes'] / max(1, ps['launches'])))

    # ---- second half of the metric: optimizer steps/s of the Sella API on the model PES of
    # f(x) = 1/2 x^T A x + c/3 sum_j (u_j.x)^3 (gradient = one device matvec), order-1
    # search, default saddle settings, Davidson re-diagonalisations through the calculator boundary.
    opt_stats = None
    if not args.no_optimizer:
        from sella_amd import device as _dev
        from sella_amd.atoms import Atoms, QuadraticCubicModel
        from sella_amd.internal import Constraints
        from sella_amd.optimize.optimize import Sella
        _dev._default = ctx
        rng = np.random.RandomState(100 + rank)
        U = rng.normal(size=(8, n))
        U /= np.linalg.norm(U, axis=1)[:, None]
        atoms = Atoms(['X'] * (n // 3), 0.05 * rng.normal(size=(n // 3, 3)), pbc=True)
        atoms.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA)
        x_init = atoms.positions.copy()
        opt = Sella(atoms, order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', logfile=None,
                    constraints=Constraints(atoms), proj_trans=False)
        opt.run(fmax=0.0, steps=2)                       # warm-up incl. the initial diagonalisation
        ctx.sync()
        ncalls0 = atoms.calc.ncalls
        ts = time.perf_counter()
        nst = args.opt_steps
        fused0 = opt.fused_steps
        opt.run(fmax=0.0, steps=nst)
        ctx.sync()
        topt = time.perf_counter() - ts
        opt_stats = dict(optimizer_steps_per_s=round(nst / topt, 3), steps=nst, ms_per_step=round(1e3 * topt / nst, 2),
                         force_calls=int(atoms.calc.ncalls - ncalls0), rs='tr', method='prfo', order=1,
                         one_call_steps=int(opt.fused_steps - fused0))
        # the same search through the library loop (sella_amd/search.py: `Sella.run` inside the library for this
        # configuration — what the ensemble members below run): steps/s after the same 2-step warm-up
        try:
            from sella_amd.search import LibrarySearch
            at2 = Atoms(['X'] * (n // 3), x_init.copy(), pbc=True)
            at2.calc = QuadraticCubicModel(lambda x: ctx.symm_mm(dA, x), U, c=0.05, device_matrix=dA)
            kw2 = dict(order=1, eta=1e-4, gamma=0.1, delta0=0.1, rs='tr', constraints=Constraints(at2), proj_trans=False)
            if LibrarySearch.applies(at2, **kw2):
                ls = LibrarySearch(at2, **kw2)
                ls.run(0.0, 2)
                ctx.sync()
                tl = time.perf_counter()
                ls.run(0.0, nst)
                ctx.sync()
                tl = time.perf_counter() - tl
                opt_stats['library_loop'] = dict(optimizer_steps_per_s=round(nst / tl, 3), ms_per_step=round(1e3 * tl / nst, 3),
                                                 steps=nst, one_call_steps=int(ls.one_call_steps))
                # roofline of the optimizer step: its one n x n pass is the force matvec of the model PES (8 n^2 bytes);
                # everything else is O(n r) on the structured Hessian B = lam0 I + W^T (mu - lam0) W: 8 passes over the
                # r x n eigenvector panel (dots, two Gram-Schmidt sweeps, new rows, clean-up, W+ = Q^T E read + write,
                # g_perp), the step itself one more.  Instrumented steps (hipEvents on the kernels' own packets).
                ctx.prof_reset()
                ctx.prof_enable(True)
                nprof_o = 10
                tp = time.perf_counter()
                ls.run(0.0, nprof_o)
                ctx.sync()
                tp = time.perf_counter() - tp
                ctx.prof_enable(False)
                pm = ctx.prof_get(0)
                r_now = int(ls.rank)
                alg = 8.0 * n * n + 9.0 * 8.0 * n * max(1, r_now)
                step_s = tl / nst
                if roof is not None and pm['launches'] > 0 and pm['ms'] > 0:
                    mv = hbm(pm, 'gemv_rows_kernel<1,2> (force matvec A x of the model PES, the one n x n pass of a step)')
                    roof['optimizer_step'] = dict(
                        bound='hbm', dominant_kernel=mv, explicit_rank=r_now,
                        algorithmic_bytes_per_step=round(alg), achieved=round(alg / step_s / 1e9, 1), peak=HBM_PEAK_GBS,
                        unit='GB/s', frac=round(alg / step_s / 1e9 / HBM_PEAK_GBS, 4),
                        ms_per_step=round(1e3 * step_s, 3), ms_per_step_instrumented=round(1e3 * tp / nprof_o, 3),
                        note='latency bound: ~25 dependent launches of 4-25 us and two host synchronisations per step; '
                             'kernel timeline in profiles/r05_opt_step_timeline.txt')
                ls.close()
        except Exception as e:                           # noqa: BLE001 — reported, the leg above stands on its own
            opt_stats['library_loop'] = dict(error=str(e)[:200])
        # ---- ensemble (BASELINE configs[3]): independent 256-atom-equivalent searches (3N = 768),
        # 8 per GPU, sharded round-robin over the ranks, one all-gather of the summaries at the end
        if args.ensemble_per_gpu > 0:
            from sella_amd.ensemble import local_members, run_ensemble
            ne, total = args.ensemble_n, args.ensemble_per_gpu * world
            # host-side data of this rank's members is prepared outside the timed region; the device upload
            # happens inside it, in the process (or thread) that runs the member.  --ensemble-procs P > 1: the
            # members run in P worker processes on this GPU (sella_amd.ensemble.EnsemblePool), started before the clock
            # like the library and the context of this process.
            from sella_amd.ensemble import EnsemblePool
            make_member = EnsembleMember(ne)
            mine_e = local_members(total, rank, world)
            # Since the searches run inside the library (sella_amd/search.py: no interpreter between the force calls)
            # host THREADS of this one process share the GPU — one persistent device context each — and that is the
            # default; --ensemble-procs P > 1 with --ensemble-threads 0 selects the worker processes of round 2.
            from sella_amd.ensemble import EnsembleThreads
            cpus_rank = max(1, effective_cpu_count() // max(1, local_world))
            nthr_e = args.ensemble_threads
            if nthr_e < 0:
                nthr_e = min(args.ensemble_per_gpu, 8, 2 * cpus_rank)
            nproc_e = args.ensemble_procs
            if nproc_e < 0:
                nproc_e = 0 if nthr_e > 1 else min(args.ensemble_per_gpu, EnsemblePool.BEST_PER_GPU, cpus_rank)
            pool, pool_note, tpool = None, None, None
            if nthr_e > 1 and nproc_e <= 1:
                tpool = EnsembleThreads(nthr_e)
                tpool.prepare(make_member, mine_e)
            if tpool is None and nproc_e > 1 and os.environ.get('SELLA_BENCH_COMM') != 'gloo':
                try:
                    pool = EnsemblePool(nproc_e)
                    pool.prepare(make_member, mine_e)
                except Exception as e:                   # noqa: BLE001 — the leg then runs in this process, and says so
                    pool_note = 'worker pool unavailable (%s): members run in the rank process' % (str(e)[:200],)
                    sys.stderr.write('[bench rank %d] %s\n' % (rank, pool_note))
                    if pool is not None:
                        pool.close()
                    pool = None
            if pool is None and tpool is None:
                for i in mine_e:
                    make_member.prepare(i)
            # the leg is tens of milliseconds long (8 members, one per thread: the slowest thread's wake-up is in it), so
            # it is run `--ensemble-reps` times — every pass complete: members built, uploaded, searched, gathered —
            # and the MEDIAN pass is reported, all of them listed beside it
            passes = []
            for _ in range(max(1, args.ensemble_reps)):
                barrier()
                te = time.perf_counter()
                res = run_ensemble(make_member, total, fmax=0.0, steps=args.ensemble_steps,
                                   sella_kwargs=EnsembleMember.SELLA_KW,
                                   threads=tpool if tpool is not None else 1, pool=pool, prepared=pool is not None)
                ctx.sync()
                passes.append(comm.max_host(time.perf_counter() - te))
            tens = sorted(passes)[len(passes) // 2]
            nst_tot = float(res['summary'][:, 1].sum())
            opt_stats['ensemble'] = dict(replicas=total, per_gpu=args.ensemble_per_gpu, n=ne,
                                         host_threads_per_gpu=(tpool.threads if tpool is not None else 1),
                                         searches_in_library=bool(__import__('sella_amd.ensemble').ensemble.USE_LIBRARY_SEARCH),
                                         worker_processes_per_gpu=(pool.processes if pool is not None else 0),
                                         steps_per_replica=args.ensemble_steps,
                                         optimizer_steps_per_s=round(nst_tot / tens, 2),
                                         searches_per_s=round(total / tens, 3), seconds=round(tens, 3),
                                         passes_seconds=[round(t, 3) for t in passes], reported='median pass',
                                         lambda_min_negative=int((res['summary'][:, 4] < 0).sum()))
            if tpool is not None and world == 1 and args.ensemble_saturate > 0:
                # the same members with more of them in flight than the configuration names (8 per GPU): what one
                # process sustains once every thread always has a next member to take
                nsat, tsat = args.ensemble_saturate * args.ensemble_per_gpu, min(12, 2 * cpus_rank)
                with EnsembleThreads(tsat) as sat:
                    sat.prepare(make_member, range(nsat))
                    run_ensemble(make_member, tsat, fmax=0.0, steps=3, sella_kwargs=EnsembleMember.SELLA_KW, threads=sat)
                    ts0 = time.perf_counter()
                    rs = run_ensemble(make_member, nsat, fmax=0.0, steps=args.ensemble_steps,
                                      sella_kwargs=EnsembleMember.SELLA_KW, threads=sat)
                    tsat_s = time.perf_counter() - ts0
                opt_stats['ensemble']['saturated'] = dict(replicas=nsat, host_threads_per_gpu=tsat, worker_processes_per_gpu=0,
                                                          searches_per_s=round(nsat / tsat_s, 3), seconds=round(tsat_s, 3),
                                                          lambda_min_negative=int((rs['summary'][:, 4] < 0).sum()))
            if tpool is not None and world == 1 and args.ensemble_emt > 0:
                # configs[3] as named: 256-atom EMT slab members (pinned lower half, 'ras'), same threads
                emt_member = EmtSlabMember()
                tpool.prepare(emt_member)
                run_ensemble(emt_member, tpool.threads, fmax=0.0, steps=3, sella_kwargs=EmtSlabMember.SELLA_KW, threads=tpool)
                epasses = []
                for _ in range(max(1, args.ensemble_reps)):
                    for i_ in range(args.ensemble_emt):
                        emt_member.prepare(i_)
                    t0e = time.perf_counter()
                    re_ = run_ensemble(emt_member, args.ensemble_emt, fmax=0.0, steps=args.ensemble_steps,
                                       sella_kwargs=EmtSlabMember.SELLA_KW, threads=tpool)
                    epasses.append(time.perf_counter() - t0e)
                te_ = sorted(epasses)[len(epasses) // 2]
                opt_stats['ensemble']['emt_members'] = dict(replicas=args.ensemble_emt, atoms=256, n=768, nfree=384,
                                                            host_threads_per_gpu=tpool.threads, worker_processes_per_gpu=0,
                                                            steps_per_replica=args.ensemble_steps,
                                                            searches_per_s=round(args.ensemble_emt / te_, 3),
                                                            optimizer_steps_per_s=round(float(re_['summary'][:, 1].sum()) / te_, 2),
                                                            seconds=round(te_, 3),
                                                            passes_seconds=[round(t, 3) for t in epasses],
                                                            lambda_min_negative=int((re_['summary'][:, 4] < 0).sum()))
                # ... and the same members in lockstep COHORTS (sella_amd.ensemble.EnsembleCohorts -> csrc/cohort.hip: the
                # replica dimension in the kernels — one batched launch per kernel of the step for all members of a cohort,
                # one stream synchronisation per phase): the 8 members of one GPU's share, then 8 x as many in flight
                from sella_amd.ensemble import EnsembleCohorts
                co_stats = {}
                # (member threads — the members' host code on a worker thread each, width + 1 spinning cores per cohort —
                #  for the GPU's own share when the host has the cores; fibers for the saturated case)
                for tag, nmem_c, width_c, thr_c, mt_c in (('share_of_one_gpu', args.ensemble_emt, max(1, args.ensemble_emt // 2), 2, cpus_rank >= 12),
                                                          ('saturated', 8 * args.ensemble_emt, 8, min(8, 2 * cpus_rank), False)):
                    with EnsembleCohorts(width_c, thr_c, member_threads=mt_c) as cohorts:
                        cohorts.prepare(emt_member)
                        run_ensemble(lambda i: emt_member(-1 - i), width_c * thr_c, fmax=0.0, steps=3,
                                     sella_kwargs=EmtSlabMember.SELLA_KW, cohort=cohorts)
                        cpasses, before = [], cohorts.stats()
                        for _ in range(max(1, args.ensemble_reps)):
                            for i_ in range(nmem_c):
                                emt_member.prepare(i_)
                            t0c = time.perf_counter()
                            rc_ = run_ensemble(emt_member, nmem_c, fmax=0.0, steps=args.ensemble_steps,
                                               sella_kwargs=EmtSlabMember.SELLA_KW, cohort=cohorts)
                            cpasses.append(time.perf_counter() - t0c)
                        after = cohorts.stats()
                    tc_ = sorted(cpasses)[len(cpasses) // 2]
                    npass = len(cpasses)
                    co_stats[tag] = dict(replicas=nmem_c, cohort_width=width_c, issuing_threads=thr_c, member_threads=bool(mt_c),
                                         searches_per_s=round(nmem_c / tc_, 3), seconds=round(tc_, 3),
                                         passes_seconds=[round(t, 3) for t in cpasses],
                                         launches_asked_per_pass=(after['launches_asked'] - before['launches_asked']) // npass,
                                         launches_issued_per_pass=(after['launches_issued'] - before['launches_issued']) // npass,
                                         waits_asked_per_pass=(after['waits_asked'] - before['waits_asked']) // npass,
                                         stream_syncs_per_pass=(after['stream_syncs'] - before['stream_syncs']) // npass,
                                         bit_identical_to_threads=bool(np.array_equal(rc_['summary'][:args.ensemble_emt], re_['summary'])),
                                         lambda_min_negative=int((rc_['summary'][:, 4] < 0).sum()))
                opt_stats['ensemble']['emt_members']['cohorts'] = co_stats
            if pool_note:
                opt_stats['ensemble']['note'] = pool_note
            if pool is not None:
                pool.close()
            if tpool is not None:
                tpool.close()
            settle()
        # ---- BASELINE configs[1] as named: 1024-atom Cu(111) EMT slab (3N = 3072), one surface atom lifted onto a
        # bridge site, lower half frozen by translation constraints (the README pattern), default Sella settings,
        # device EMT calculator.  Host-glue bound (Python between sub-millisecond kernels), reported for the record.
        if args.emt_steps > 0 and world == 1:
            from sella_amd.atoms import EMT
            from tools.emt_slab_opt import make_slab
            slab = make_slab()
            cons_s = Constraints(slab)
            for atom in slab:
                if atom.position[2] < slab.cell[2, 2] / 2.:
                    cons_s.fix_translation(atom.index)
            slab.calc = EMT()
            dyn = Sella(slab, constraints=cons_s, logfile=None)
            dyn.run(0.0, 2)
            ctx.sync()
            nc0 = slab.calc.ncalls
            fs0 = dyn.fused_steps
            tse = time.perf_counter()
            dyn.run(0.0, args.emt_steps)
            ctx.sync()
            tsl = time.perf_counter() - tse
            opt_stats['emt_slab'] = dict(atoms=len(slab), n=3 * len(slab), nfree=int(dyn.pes.get_Ufree().shape[1]),
                                         steps=args.emt_steps, optimizer_steps_per_s=round(args.emt_steps / tsl, 2),
                                         ms_per_step=round(1e3 * tsl / args.emt_steps, 1),
                                         force_calls=int(slab.calc.ncalls - nc0), rs='ras', calculator='EMT (device)',
                                         one_call_steps=int(dyn.fused_steps - fs0))
            try:
                from sella_amd.search import LibrarySearch
                slab2 = make_slab()
                cons2 = Constraints(slab2)
                for atom in slab2:
                    if atom.position[2] < slab2.cell[2, 2] / 2.:
                        cons2.fix_translation(atom.index)
                slab2.calc = EMT()
                if LibrarySearch.applies(slab2, constraints=cons2):
                    ls = LibrarySearch(slab2, constraints=cons2)
                    ls.run(0.0, 2)
                    ctx.sync()
                    tl = time.perf_counter()
                    ls.run(0.0, args.emt_steps)
                    ctx.sync()
                    tl = time.perf_counter() - tl
                    opt_stats['emt_slab']['library_loop'] = dict(optimizer_steps_per_s=round(args.emt_steps / tl, 2),
                                                                 ms_per_step=round(1e3 * tl / args.emt_steps, 3),
                                                                 force_calls=int(ls.neval))
                    # roofline of the step on the configuration BASELINE's metric names.  There is no n x n pass in it: the
                    # approximate Hessian is lam0 I + W^T (mu - lam0) W with r explicit pairs (and its principal submatrix on
                    # the free coordinates, r_view pairs), so a step is O(n r): 9 passes over each of the two panels (dots, two
                    # Gram-Schmidt sweeps, new rows, clean-up, W+ = Q^T E read + write, g_perp), the panel products of the
                    # root search over the view (3.4 batches of 16 trial steps on average, rocprofv3) and one EMT force call
                    # (positions in, forces out, 9 KB of neighbour hand-over per atom written and read).
                    r_f, r_v = int(ls.rank), max(0, int(ls.rank_view))
                    n_s, m_s, na_s = 3 * len(slab2), int(dyn.pes.get_Ufree().shape[1]), len(slab2)
                    alg_s = 9.0 * 8.0 * n_s * max(1, r_f) + 9.0 * 8.0 * m_s * max(1, r_v) + 3.4 * 8.0 * m_s * (r_v + 2) \
                        + 48.0 * na_s + 2.0 * 9216.0 * na_s
                    step_slab = tl / args.emt_steps
                    if roof is not None:
                        roof['optimizer_step_emt_slab'] = dict(
                            bound='hbm', explicit_rank=r_f, explicit_rank_view=r_v, algorithmic_bytes_per_step=round(alg_s),
                            achieved=round(alg_s / step_slab / 1e9, 2), peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=round(alg_s / step_slab / 1e9 / HBM_PEAK_GBS, 5), ms_per_step=round(1e3 * step_slab, 3),
                            note='latency bound: ~63 dependent launches of 3-30 us (two coordinate-update chains, 3.4 round trips '
                                 'of the per-atom root search, EMT) and their host waits; kernel timeline in '
                                 'profiles/r05_emt_step_timeline.txt')
                    ls.close()
            except Exception as e:                       # noqa: BLE001
                opt_stats['emt_slab']['library_loop'] = dict(error=str(e)[:200])
        _dev._default = None

    # ---- BASELINE configs[4]: block Davidson, 16 new vectors per iteration, H.V panel on the matrix cores, rows of
    # H sharded over the ranks (strong scaling: the operator is fixed, every rank streams n / N rows of it and ONE
    # all-gather per block iteration assembles H V).  The operator is a dense symmetric pseudo-random matrix built
    # panel by panel from an integer hash (no rank ever holds more than its rows); the iteration count is fixed,
    # so the figure is time per block iteration, not convergence (that is tests/test_big_gpu.py).  The tolerance is the
    # parity tests' (1e-10: no pair reaches it in 12 iterations, residuals are ~0.1); rounds 1-5 passed 1e-14, which no
    # path can meet and which makes the pipelined driver (round 6) hold A T to 4 eps |A| by taking its matrix pass late.
    block_stats = None
    if args.block_n > 0 and args.block_iters > 0:
        from sella_amd import device as _dev2
        from sella_amd.parallel import RowShardedOperator
        _dev2._default = ctx
        nb_ = args.block_n
        m_ = -(-nb_ // world)
        lo_, hi_ = min(rank * m_, nb_), min((rank + 1) * m_, nb_)
        ii = np.arange(lo_, hi_, dtype=np.int64)[:, None]
        jj = np.arange(nb_, dtype=np.int64)[None, :]
        a_, b_ = np.minimum(ii, jj), np.maximum(ii, jj)
        hsh = ((a_ * 73856093) ^ (b_ * 19349663) ^ 0x5bd1e995) & 0xFFFFF
        Hloc = (hsh.astype(np.float64) / 0xFFFFF - 0.5) * 0.02
        Hloc[np.arange(hi_ - lo_), np.arange(lo_, hi_)] += 0.5 + 50.0 * (np.arange(lo_, hi_) / nb_) ** 2
        dgl = np.zeros(nb_)
        dgl[lo_:hi_] = Hloc[np.arange(hi_ - lo_), np.arange(lo_, hi_)]
        dg_all = comm.allgather_host(dgl).sum(axis=0) if world > 1 else dgl
        op = RowShardedOperator(Hloc, lo_, nb_)
        del Hloc, hsh, a_, b_
        # (maxvec = 48: the basis limit of rounds 1 - 5, one block of room, so that the per-iteration figure stays comparable; the
        #  library's default is 64 since round 6 — half the iterations and 0.58 of the time to convergence,
        #  tools/block_restart_sweep.py — at 0.48 instead of 0.42 ms per iteration there)
        op.block_davidson(16, block=16, tol=1e-10, maxiter=2, maxvec=48, diag=dg_all)          # warm-up
        # one panel pass alone (the roofline-relevant part of the iteration): 8 * rows * n bytes
        Xp = np.random.RandomState(1).standard_normal((nb_, 16))
        op.local_matmat(Xp)
        ctx.prof_reset()
        ctx.prof_enable(True)
        for _ in range(3):
            op.local_matmat(Xp)
        ctx.prof_enable(False)
        pp = ctx.prof_get(0)
        barrier()
        tb = time.perf_counter()
        outb = op.block_davidson(16, block=16, tol=1e-10, maxiter=args.block_iters, maxvec=48, diag=dg_all)
        ctx.sync()
        tblk = comm.max_host(time.perf_counter() - tb)
        nit = max(1, outb['niter'])
        block_stats = dict(n=nb_, block=16, rows_per_rank=int(hi_ - lo_), iterations=int(outb['niter']),
                           ms_per_block_iter=round(1e3 * tblk / nit, 3), block_iter_per_s=round(nit / tblk, 2),
                           vectors_per_s=round(outb['nmatvec'] / tblk, 1),
                           panel_pass_us=round(1e3 * pp['ms'] / max(1, pp['launches']), 1),
                           panel_pass_gbs=round(pp['bytes'] / max(1e-12, pp['ms'] * 1e-3) / 1e9, 1) if pp['launches'] else None,
                           lowest_ritz=float(outb['lams'][0]), scaling='strong (rows of H sharded over the ranks)',
                           preconditioner='diagonal')
        # time to convergence with the library's defaults (basis limit nev + 3 block): the figure a caller sees
        try:
            barrier()
            tcv0 = time.perf_counter()
            outc = op.block_davidson(16, block=16, tol=1e-9, maxiter=600, diag=dg_all)
            ctx.sync()
            tcv = comm.max_host(time.perf_counter() - tcv0)
            block_stats['converged_run'] = dict(tol=1e-9, pairs=int(outc['nconv']), iterations=int(outc['niter']),
                                                products=int(outc['nmatvec']), ms=round(1e3 * tcv, 1),
                                                basis_limit='default (nev + 3 block = 64)')
        except Exception as e:                           # noqa: BLE001
            block_stats['converged_run'] = dict(error=str(e)[:200])
        if pp['launches']:
            # Amdahl: only the panel pass shards over the ranks (rows of H); the rest of a block iteration is replicated
            t_it, t_pp = 1e3 * tblk / nit, 1e-3 * block_stats['panel_pass_us'] * world
            ser = max(0.0, 1.0 - t_pp / t_it) if world == 1 else None
            if ser is not None:
                block_stats['serial_fraction'] = round(ser, 3)
                block_stats['amdahl_speedup_at_8_ranks'] = round(1.0 / (ser + (1.0 - ser) / 8.0), 2)
        if pp['launches']:
            # the one MFMA kernel of the path: 2 * rows * n * 16 flop per pass against the dense fp64 MFMA peak
            # (78.6 TFLOP/s, MI355X_MICROARCH.md); it is an HBM-bound product — 16 flop per 8 streamed bytes
            tfl = 2.0 * (hi_ - lo_) * nb_ * 16 / max(1e-12, pp['ms'] * 1e-3 / pp['launches']) / 1e12
            block_stats.update(panel_pass_tflops=round(tfl, 2), panel_pass_mfma_frac=round(tfl / 78.6, 3),
                               panel_pass_hbm_frac=round(block_stats['panel_pass_gbs'] / HBM_PEAK_GBS, 3))
        if world == 1 and args.block_eigh:
            # the `exact` diagonalisation configs[4] needs once per call with an eigenbasis preconditioner
            # (sella/eigensolvers.py:9-28): device eigh of the resident 3N x 3N operator, second call timed
            try:
                for rep in range(2):
                    te0 = time.perf_counter()
                    w_, V_, Vt_ = ctx.eigh(op.dH)
                    ctx.sync()
                    t_e = time.perf_counter() - te0
                    V_.free()
                    Vt_.free()
                block_stats['eigh_exact_ms'] = round(1e3 * t_e, 1)
                block_stats['eigh_exact_lowest'] = float(w_[0])
            except Exception as e:                               # noqa: BLE001 — the figure is optional, the line is not
                block_stats['eigh_exact_note'] = str(e)[:120]
        _dev2._default = None

    times = [elapsed]
    total_iters = iters
    if world > 1:
        gathered = comm.allgather_host(np.array([elapsed, float(iters)]))     # RCCL all-gather of the per-replica results
        times = [float(x[0]) for x in gathered]
        total_iters = int(sum(float(x[1]) for x in gathered))
    tmax = max(times)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle.sella_oracle as orc        # checker / baseline only (oracle/README.md)
        threads = os.cpu_count()
        try:
            from threadpoolctl import threadpool_info
            info = threadpool_info()
            if info:
                threads = info[0].get('num_threads', threads)
        except Exception:
            pass
        tc = time.perf_counter()
        lc, Vc, AVc = orc.rayleigh_ritz(A, args.gamma, P, v0=g, method='jd0', maxiter=args.maxiter)
        tcpu = time.perf_counter() - tc
        # Step-for-step parity where it is defined: the Krylov trajectory amplifies a 1-ulp change about
        # 10x per iteration (tools/krylov_sensitivity.py), so the comparison that means
        # something is the Ritz value after the first few expansions, not after 25-30 of them.
        t4 = time.perf_counter()
        l4c, _, _ = orc.rayleigh_ritz(A, args.gamma, P, v0=g, method='jd0', maxiter=4)
        t4 = time.perf_counter() - t4
        l4h, _, _, _ = ctx.davidson(dA, n, g, args.gamma, method='jd0', maxiter=4, Pvecs=V, PvecsT=Vt, pevals=w)
        # the same 4-iteration sample on ONE core (SURVEY.md section 8d asks for both ends)
        one_core = None
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=1):
                t1c = time.perf_counter()
                orc.rayleigh_ritz(A, args.gamma, P, v0=g, method='jd0', maxiter=4)
                t1c = time.perf_counter() - t1c
            one_core = dict(value=round(4 / t1c, 4), cores=1, sample=f'4 iterations, {t1c:.1f} s '
                            f'(the same 4 iterations on {int(threads)} threads: {t4:.1f} s)')
        except ImportError:
            pass
        cpu = dict(value=round(Vc.shape[1] / tcpu, 4), unit='davidson_iter/s', cores=int(threads), kind='port',
                   sample=f'1 call of oracle rayleigh_ritz (reference algorithm: dense LU per iteration), '
                          f'n={n}, k={Vc.shape[1]} vectors, {tcpu:.1f} s',
                   k=int(Vc.shape[1]), one_core=one_core,
                   lam0_rel_diff_after_4_iterations=float(abs(l4c[0] - l4h[0]) / abs(l4c[0])),
                   lam0_abs_diff_at_exit=float(abs(lc[0] - lams[0])),
                   note='the exit point of the gamma = 0.1 run is chaotic in the reference itself (DESIGN.md section 4): '
                        'compare after 4 iterations, and the converged run in parity.converged_run')

    if rank == 0 and roof is not None:
        # the whole timed step in SURVEY section 8(d)'s own unit: 24 n^2 bytes per Davidson vector (A t, Q^T [r v],
        # Q [a b]) + 8 n^3 / 3 bytes of trailing-matrix reads of the one-stage tridiagonalisation of P, per call
        vec_per_call = total_iters / (args.steps * world)
        alg = 24.0 * n * n * vec_per_call + 8.0 * n ** 3 / 3.0
        gbs = alg / (tmax / args.steps) / 1e9
        roof['whole_step'] = dict(algorithmic_bytes=round(alg), davidson_bytes=round(24.0 * n * n * vec_per_call),
                                  eigh_bytes=round(8.0 * n ** 3 / 3.0), achieved_gbs=round(gbs, 1),
                                  frac=round(gbs / HBM_PEAK_GBS, 4),
                                  loop_only_gbs=round(24.0 * n * n * (it2 / t_loop) / 1e9, 1),
                                  loop_only_frac=round(24.0 * n * n * (it2 / t_loop) / 1e9 / HBM_PEAK_GBS, 4))
    if rank == 0:
        value = total_iters / tmax
        line = {
            'metric': 'Davidson iterations/s (3N=3072 synthetic Hessian, jd0, gamma=0.1, maxiter=40; '
                      'whole rayleigh_ritz call incl. device eigh of the preconditioner)',
            'value': round(value, 2), 'unit': 'davidson_iter/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * tmax / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': f'1024-atom-equivalent 3N={n} fp64 Davidson (BASELINE configs[1]), '
                                   f'one independent replica per GPU', 'n': n, 'maxiter': args.maxiter,
                       'gamma': args.gamma, 'method': 'jd0', 'problems': args.seeds * args.starts, 'matrices': args.seeds,
                       'vectors_per_call': round(total_iters / (args.steps * world), 2),
                       **({'options': list(args.option)} if args.option else {})},
            'davidson_loop_only_iter_per_s': round(it2 / t_loop, 1),
            'ms_per_vector_loop': round(1e3 * t_loop / max(1, it2), 4),
            'eigh_ms': round(1e3 * t_eigh, 2),
            'upload_ms': round(upload_ms, 2),          # A and P from numpy memory (2 x 75 MB at 3N = 3072): NOT in `value`
            'concurrent_problems': concurrent,
            'optimizer': opt_stats,
            'block_davidson': block_stats,
            'parity': {'lowest_ritz_value': float(lams[0]), 'ritz_residual_norm': resid, 'max_abs_AV_minus_A_V': av_err,
                       'converged_run': conv},
            'collective': {'kind': comm.kind, 'nranks': getattr(comm, 'nranks', comm.world)},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(line), flush=True)
    comm.barrier()
    comm.close()
    if comm.kind == 'torch.distributed':
        comm.dist.destroy_process_group()


if __name__ == '__main__':
    main()
