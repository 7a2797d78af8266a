#!/usr/bin/env python3
"""Headline benchmark: CGNR iterations/sec @ 4096x2048 ComplexF32 (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

N = 1: a step is ONE CGNR iteration (src/CGNR.jl:143-178) of the matrix-free normal operator on a dense
column-major ComplexF32 4096x2048 A that is already resident in HBM (lambda = 0, relTol = 0, SURVEY.md 8d
"headline metric run").  The timed region is EXACTLY K iterations between barrier + synchronize on both sides;
when that region is shorter than 10 ms it is repeated (>= 50 times, every repetition bracketed the same way) and
`value` = K / median(elapsed), with the spread reported beside it.

N > 1: one process per GPU.  `python bench.py --gpus N` from a plain shell starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` itself, as a CHILD process, before anything has
touched torch or HIP, relays its output and exit code (`--dry-launch` prints that command line and exits); under an external
torchrun (RANK / WORLD_SIZE set) it runs as a rank.  The SAME workload and unit on every GPU -- one
independent headline solve per GPU (the path shards by independent problems, src/MultiThreading.jl:30-79: no data-path
collective, weak scaling), `value` = iterations/s summed over the GPUs, so value(N) / (N * value(1)) is an efficiency.
Every N > 1 line also carries `n1_same_workload_value` (rank 0 running the same steps alone while the other ranks wait)
and the quotient, plus `config4_batched`: BASELINE configs[3]'s shared-A flavour (8 right-hand sides per GPU advancing
together on the matrix cores, solve-iterations/s) measured the same way.  `--workload config4` makes that the line's
workload; `--workload rowsharded` runs BASELINE configs[4] (one tall A row-partitioned, one all-reduce per iteration).

Prints ONE JSON line on rank 0 with `roofline` and (N = 1) `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_KIND = ("static: PMC counters cannot be read from inside this process; the figure is the committed `rocprofv3 --pmc` pass named in "
                "traffic_source (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, separate passes), per launch of the same kernel and shape")
HBM_PEAK_GBS = 8000.0     # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md chip table
MFMA_F32_PEAK_TF = 157.3  # dense f32-input MFMA peak, same table
VALU_F32_PEAK_TF = 157.3  # "Peak FP32 (vector)", same table: 64 FLOP/clk/SIMD, packed FMAs included (round 3 used twice that)
L2_SHARED_TBS = 16.8      # rows shared by every workgroup of an XCD, out of its L2 (guide, L2 section: 16.8-18.8 TB/s)
INF_CACHE_TBS = 8.6       # Infinity Cache (guide, same table): what crosses XCDs travels at this rate at best
# CG converges geometrically on the well-conditioned randn matrix (hard part 4): left running for hundreds
# of iterations the recursive residual underflows Float32 and alpha becomes 0/0.  The bench therefore times
# back-to-back SOLVES of SEGMENT iterations each (BASELINE configs 4/5 use 32); the init! of every solve after the
# first (one extra GEMV) is inside the timed region but not counted as a step.
SEGMENT = 32
PATH_NAMES = {0: "two GEMVs + update kernel", 1: "one-pass slab pipeline (2 launches per iteration)",
              2: "Gram-mode pipeline", 3: "batched matrix-core kernels", 4: "resident (one launch per step call, A in registers)",
              5: "resident Gram mode (one launch per step call, AHA in registers)"}


def grid_exchange_floor():
    """the resident kernels' in-kernel exchange measured with the arithmetic stripped (tools/ubench/grid_barrier, built by
    __graft_entry__.build()): us per round of the bare grid barrier, of the flat two-hop all-reduce and of the two-level
    one.  Run live (about a second); None when the binary is missing."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "grid_barrier")
    if not os.path.exists(exe):
        return None
    try:
        txt = subprocess.run([exe, "1000", "2048"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return None
    best = {}
    for line in txt.splitlines():
        f = line.split()
        if len(f) >= 2 and f[0] in ("bar", "flat", "gbar") and "fail 0" in line and "wrong sums 0" in line:
            best[f[0]] = min(best.get(f[0], 1e9), float(f[1]))
        if line.startswith("group GN= 8 strided") and "fail 0" in line and "wrong sums 0" in line:
            best["two_level"] = min(best.get("two_level", 1e9), float(f[4]))
        if len(f) >= 2 and f[0] == "groupl2" and "fail 0" in line and "wrong sums 0" in line and "misplaced workgroups 0" in line:
            best["two_level_l2_rows"] = min(best.get("two_level_l2_rows", 1e9), float(f[1]))
    return best or None


def make_A(M, N, seed, dtype=np.complex64):
    """zero-mean normal entries, complex = (g1 + i g2)/sqrt 2; generated in float32"""
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "c":
        A = np.empty((M, N), dtype=np.complex64, order="F")
        s = np.float32(1 / math.sqrt(2))
        A.real = rng.standard_normal((N, M), dtype=np.float32).T * s
        A.imag = rng.standard_normal((N, M), dtype=np.float32).T * s
    else:
        A = np.asfortranarray(rng.standard_normal((N, M), dtype=np.float32).T)
    return A


def bytes_per_cgnr_iteration(M, N, s):
    """algorithmic bytes: A read twice + the length-N / length-M vector traffic"""
    return 2 * M * N * s + (16 * N + 2 * M) * s


def float64_cgnr(A, b, n):
    """the in-band proof that the timed kernel did the work: n iterations of CG on the normal equations in complex128 on the host
    (src/CGNR.jl:107-178 with lambda = 0: r = A'b, p = r; alpha = |r|^2 / <p, A'A p>, x += alpha p, r -= alpha A'A p,
    beta = |r_new|^2 / |r|^2, p = beta p + r) -- a few lines of NumPy inside the measurement script, NOT the oracle package, so
    that the bench line can vouch for its own `value` wherever it runs"""
    A64 = A.astype(np.complex128)
    AH = np.ascontiguousarray(A64.conj().T)
    x = np.zeros(A.shape[1], np.complex128)
    r = AH @ b.astype(np.complex128)
    p = r.copy()
    for _ in range(n):
        v = AH @ (A64 @ p)
        zeta = np.vdot(r, r).real
        alpha = zeta / np.vdot(p, v)
        x += alpha * p
        r -= alpha * v
        p = (np.vdot(r, r).real / zeta) * p + r
    return x


def spread(xs):
    xs = sorted(xs)
    n = len(xs)
    return {"n": n, "median": statistics.median(xs), "min": xs[0], "max": xs[-1], "p10": xs[int(0.1 * (n - 1))],
            "p90": xs[int(math.ceil(0.9 * (n - 1)))]}


# ---- CPU baselines (rank 0, N = 1 only): the oracle's NumPy restatement and its C++/OpenMP restatement -----------
def usable_cpus():
    """CPUs this process can really keep busy: the affinity mask capped by the cgroup CPU quota (the GPU boxes give a
    container 16 CPUs' worth of time on a 256-CPU host; more threads than that run in bursts and are then throttled,
    which a short calibration does not see)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline_numpy(A, b, budget_s=8.0, max_iters=1 << 30):  # (time-bounded: whole solves until budget_s is spent)
    """the oracle's CGNR (NumPy/OpenBLAS restatement of src/CGNR.jl:143-178) timed on the host.  OpenBLAS's cgemv does
    not scale to every core of a big host, so a short calibration picks the BLAS thread count (reported as `cores`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rls_oracle as O

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None

    def run(n_solves, limit=None):
        s = O.CGNR(A, iterations=SEGMENT, relTol=0.0)
        t0 = time.perf_counter()
        done = 0
        while done < n_solves and (limit is None or time.perf_counter() - t0 < limit):
            s.init(b)  # same cadence as the GPU leg: one solve = init! + SEGMENT iterations
            for _ in range(SEGMENT):
                s.iterate()
            done += 1
        return done * SEGMENT, time.perf_counter() - t0

    ncpu, quota = usable_cpus()
    best_threads, calib = ncpu, {}
    if threadpool_limits is not None:
        for nt in sorted({1, 4, 8, 16, 32, ncpu}):
            if nt > ncpu:
                continue
            with threadpool_limits(limits=nt, user_api="blas"):
                run(1 << 20, 0.3)  # warm
                n, dt = run(1 << 20, 0.8)
            calib[nt] = n / dt
        best_threads = max(calib, key=calib.get)
        ctxmgr = threadpool_limits(limits=best_threads, user_api="blas")
    else:
        import contextlib
        ctxmgr = contextlib.nullcontext()
    with ctxmgr:
        n, dt = run(max_iters // SEGMENT, budget_s)
    return {"value": n / dt, "unit": "iterations/s", "cores": int(best_threads), "kind": "port",
            "implementation": "NumPy/OpenBLAS restatement (oracle/rls_oracle.py)",
            "sample": f"{n} CGNR iterations of the same {A.shape[0]}x{A.shape[1]} complex64 problem, {dt:.1f} s, BLAS threads chosen "
                      f"by calibration { {k: round(v, 1) for k, v in calib.items()} } it/s; {ncpu} usable CPUs (cgroup quota "
                      f"{quota}) of {os.cpu_count()} on the host",
            "GBps_algorithmic": bytes_per_cgnr_iteration(A.shape[0], A.shape[1], 8) * n / dt / 1e9, "ms_per_step": 1e3 * dt / n}


def cpu_baseline_openmp(A, b, budget_s=10.0):
    """oracle/cgnr_omp.cpp: the same iteration in C++ with OpenMP, both products streaming A once with every core
    (column panels per thread, NUMA-placed copy of A).  Thread count by calibration; checked against the NumPy
    restatement in tests/test_oracle.py."""
    so = os.path.join(ROOT, "oracle", "_build", "libcgnr_omp.so")
    if not os.path.exists(so):
        return {"error": f"{so} not built (python -c 'import __graft_entry__ as g; g.build()')"}
    lib = C.CDLL(so)
    lib.cgnr_omp_run.restype = C.c_int64
    lib.cgnr_omp_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                 C.POINTER(C.c_double)]
    M, N = A.shape
    x = np.zeros(N, np.complex64)
    sec = C.c_double()
    ncpu, quota = usable_cpus()

    def run(threads, solves):
        n = lib.cgnr_omp_run(A.ctypes.data, M, N, b.ctypes.data, SEGMENT, solves, 0.0, threads, x.ctypes.data, C.byref(sec))
        return n, sec.value

    calib = {}
    for nt in sorted({4, 8, 16, 32, 64, 128, ncpu}):
        if nt > ncpu:
            continue
        run(nt, 1)
        n, dt = run(nt, 4)
        n2, dt2 = run(nt, max(4, int(0.4 * n / dt / SEGMENT)))  # ~0.4 s: long enough to run into a CPU quota
        calib[nt] = n2 / dt2
    best = max(calib, key=calib.get)
    # the sample: chunks of about a second until the budget is used (bounded by time, not by a count sized from the
    # calibration -- a host that slows down must not stretch the bench run)
    n = 0
    dt = 0.0
    chunk = max(2, int(calib[best] / SEGMENT))
    while dt < budget_s:
        ni, di = run(best, chunk)
        n += ni
        dt += di
    return {"value": n / dt, "unit": "iterations/s", "cores": int(best), "kind": "port",
            "implementation": "C++/OpenMP restatement (oracle/cgnr_omp.cpp), x86-64-v3",
            "sample": f"{n} CGNR iterations of the same {M}x{N} complex64 problem, {dt:.1f} s, OpenMP threads chosen by calibration "
                      f"{ {k: round(v, 1) for k, v in calib.items()} } it/s; {ncpu} usable CPUs (cgroup quota {quota}) of "
                      f"{os.cpu_count()} on the host",
            "GBps_algorithmic": bytes_per_cgnr_iteration(M, N, 8) * n / dt / 1e9, "ms_per_step": 1e3 * dt / n}


def cpu_baselines(A, b):
    variants = []
    for fn in (cpu_baseline_openmp, cpu_baseline_numpy):
        try:
            variants.append(fn(A, b))
        except Exception as e:  # a broken baseline must not take the GPU line down, but it must be visible
            variants.append({"error": f"{fn.__name__}: {e!r}"})
    ok = [v for v in variants if "value" in v]
    if not ok:
        return {"value": None, "unit": "iterations/s", "cores": 0, "kind": "port", "sample": "no CPU baseline ran", "variants": variants}
    best = dict(max(ok, key=lambda v: v["value"]))
    best["variants"] = variants
    # the reference's own CPU path would be the baseline of choice if a Julia runtime with the package were
    # on this host; it is probed for, never installed (kind stays "port")
    import shutil
    jl = shutil.which("julia")
    best["reference_runtime"] = f"julia found at {jl} but RegularizedLeastSquares.jl is not installed offline" if jl else "julia not found on this host"
    return best


# ---- the other paths of SURVEY 8 on the same operator (N = 1, untimed extras) --------------------------------------
def other_paths(rls, ctx, Ad, A, b, errors):
    """a few milliseconds each, so that one bench run shows them all.  None of this enters `value`.  A failure of one
    entry is recorded in `errors` (top-level `other_paths_error` of the JSON line), it is never swallowed."""
    M, N = A.shape
    out = {}
    lib, h = ctx.lib, ctx.handle

    def timed(run, n_inner, reps=8):
        """us per inner iteration: hipEvents around each repetition, the fastest one (a host hiccup while enqueuing
        leaves the GPU idle inside the timed region; it is not the kernels')"""
        run(); run(); ctx.sync()
        best = float("inf")
        for _ in range(reps):
            ctx.timer_start()
            run()
            best = min(best, ctx.timer_stop_ms())
        return best * 1e3 / n_inner

    def entry(name):
        def deco(fn):
            try:
                out[name] = fn()
            except Exception as e:
                errors.append(f"{name}: {e!r}")
            return fn
        return deco

    rho = 0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2
    rng = np.random.default_rng(5)
    state = {}

    @entry("cgnr_two_launch_pipeline (resident = 0: A streamed once per iteration)")
    def _():
        ctx.tune(resident=0)
        try:
            S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
            rls.solve_(S, b)
            us = timed(lambda: (rls.init_(S, b), lib.rls_cgnr_step(S.state._plan, 32)), 32)
            us_a, us_r = C.c_float(), C.c_float()
            rls.init_(S, b)
            rc = lib.rls_cgnr_step_profiled(S.state._plan, 100, C.byref(us_a), C.byref(us_r))
        finally:
            ctx.tune(resident=1)
        s = 8
        res = {"us_per_iteration": us, "iterations_per_s": 1e6 / us, "algorithmic_GBps": bytes_per_cgnr_iteration(M, N, s) / us / 1e3,
               "frac_algorithmic": bytes_per_cgnr_iteration(M, N, s) / us / 1e3 / HBM_PEAK_GBS}
        if rc == 0:
            nwg = M // 16
            res["cgnr_pipe_a_kernel_us_back_to_back"] = us_a.value
            res["cgnr_pipe_a_kernel_min_hbm_bytes"] = M * N * s + nwg * N * s
            res["cgnr_pipe_a_kernel_frac_hbm"] = (M * N * s + nwg * N * s) / (us_a.value * 1e-6) / 1e9 / HBM_PEAK_GBS
            res["cgnr_pipe_r_kernel_us_back_to_back"] = us_r.value
            res["note"] = ("kernel times are back-to-back launches between two hipEvents; in the real sequence the reduce kernel reads "
                           "partial rows the slab kernel has just written on other XCDs and runs ~4.8 us (rocprofv3, profiles/)")
        return res

    @entry("iterate_per_call_cadence (the reference's solve! loop: one iterate + one `done` check per iteration, "
           "src/RegularizedLeastSquares.jl:103-117; rls_*_step_status(plan, 1) = one host synchronisation per iteration)")
    def _():
        res = {}

        def cadence(make, step_status, status_t, n_it):
            S = make()
            rls.solve_(S, b)
            st_ = status_t()
            plan = S.state._admm if hasattr(S.state, "_admm") and S.state._admm else S.state._plan
            def run():
                rls.init_(S, b)
                for _ in range(n_it):
                    step_status(plan, st_)
            run(); ctx.sync()
            best = float("inf")
            for _ in range(5):
                t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
            assert st_.iteration == n_it, (st_.iteration, n_it)
            return 1e6 * best / n_it

        L = rls._lib
        for tag, res_on in (("", 1), ("_pipeline (resident = 0)", 0)):
            ctx.tune(resident=res_on)
            try:
                us = cadence(lambda: rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0),
                             lambda p, st_: L.check(h, lib.rls_cgnr_step_status(p, 1, C.byref(st_)), "cgnr_step_status"), L.CgnrStatus, 32)
                res["cgnr" + tag] = {"us_per_iterate_call_wall": us, "iterations_per_s": 1e6 / us}
                us = cadence(lambda: rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=32, relTol=0.0),
                             lambda p, st_: L.check(h, lib.rls_fista_step_status(p, 1, C.byref(st_)), "fista_step_status"), L.FistaStatus, 32)
                res["fista_l1" + tag] = {"us_per_iterate_call_wall": us, "iterations_per_s": 1e6 / us}
                us = cadence(lambda: rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=8, iterationsCG=10,
                                                            tolInner=1e-5, absTol=0.0, relTol=0.0),
                             lambda p, st_: L.check(h, lib.rls_admm_step_status(p, 1, C.byref(st_), None, 0), "admm_step_status"), L.AdmmStatus, 8)
                res["admm_l1_outer" + tag] = {"us_per_iterate_call_wall": us, "outer_iterations_per_s": 1e6 / us}
            finally:
                ctx.tune(resident=1)
        res["note"] = ("wall clock of init! + n x (step one iteration, read the status back) driven from Python through ctypes; the headline "
                       "`value` enqueues a whole solve per call instead (solve! without callbacks)")
        return res

    @entry("fista_l1_matrix_free (BASELINE configs[1])")
    def _():
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
        rls.solve_(S, b)
        us = timed(lambda: (rls.init_(S, b), lib.rls_fista_step(S.state._plan, 48)), 48)
        return {"us_per_iteration": us, "iterations_per_s": 1e6 / us}

    @entry("optista_pogm_l1 (SURVEY 8f-1: blocks of 48 iterations as ONE resident launch, rls_pgm_step_resident)")
    def _():
        res = {}
        for name in ("OptISTA", "POGM"):
            for tag, res_on in (("resident_launches", 1), ("launch_per_iteration", 0)):
                ctx.tune(resident=res_on)
                try:
                    S = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
                    rls.solve_(S, b)
                    us = timed(lambda: (rls.init_(S, b), S._run(S.state)), 48, reps=4)
                finally:
                    ctx.tune(resident=1)
                res[f"{name}_{tag}"] = {"us_per_iteration_incl_init": us, "iterations_per_s": 1e6 / us}
        return res

    @entry("gram_gemm_AHA (setup, matrix cores)")
    def _():
        t0 = time.perf_counter(); G = Ad.gram(); ctx.sync(); t_gram = time.perf_counter() - t0
        t0 = time.perf_counter(); G = Ad.gram(); ctx.sync(); t_gram = min(t_gram, time.perf_counter() - t0)
        state["G"] = G
        state["gram_ms"] = 1e3 * t_gram
        # the kernel computes the UPPER triangle of the Hermitian result only (64 x 64 tiles, the mirror image is stored as the conjugate):
        # `TFLOPs_nominal` prices the full N x N x M product the caller gets, `TFLOPs_executed` the (N / 64)(N / 64 + 1) / 2 tiles it runs
        tiles = (N // 64) * (N // 64 + 1) // 2
        return {"ms": 1e3 * t_gram, "TFLOPs_nominal": 8.0 * N * N * M / t_gram / 1e12,
                "TFLOPs_executed": 8.0 * tiles * 64 * 64 * M / t_gram / 1e12,
                "note": "Hermitian half only: nominal counts the full A'A the caller receives and can exceed the 157.3 TF f32 MFMA peak; executed counts the tiles that run"}

    @entry("cgnr_gram_mode (AHA explicit, one launch per iteration)")
    def _():
        S = rls.createLinearSolver(rls.CGNR, Ad, AHA=state["G"], iterations=32, relTol=0.0)
        rls.solve_(S, b)
        us = timed(lambda: (rls.init_(S, b), lib.rls_cgnr_step(S.state._plan, 32)), 32)
        return {"us_per_iteration": us, "iterations_per_s": 1e6 / us}

    @entry("fista_l1_gram_mode")
    def _():
        S = rls.createLinearSolver(rls.FISTA, Ad, AHA=state["G"], reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
        rls.solve_(S, b)
        us = timed(lambda: (rls.init_(S, b), lib.rls_fista_step(S.state._plan, 48)), 48)
        return {"us_per_iteration": us, "iterations_per_s": 1e6 / us}

    @entry("iterate_per_call_cadence_gram_mode (the same loop on the reference constructors' DEFAULT operator, AHA = A' * A explicit -- "
           "src/CGNR.jl:49, src/FISTA.jl:58, src/ADMM.jl:81; the resident Gram kernels listen between calls)")
    def _():
        L = rls._lib
        res = {}

        def cadence(make, step_status, status_t, n_it):
            S = make()
            rls.solve_(S, b)
            st_ = status_t()
            plan = S.state._admm if hasattr(S.state, "_admm") and S.state._admm else S.state._plan
            def run():
                rls.init_(S, b)
                for _ in range(n_it):
                    step_status(plan, st_)
            run(); ctx.sync()
            best = float("inf")
            for _ in range(5):
                t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
            assert st_.iteration == n_it, (st_.iteration, n_it)
            return 1e6 * best / n_it

        G = state["G"]
        us = cadence(lambda: rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32, relTol=0.0),
                     lambda p, st_: L.check(h, lib.rls_cgnr_step_status(p, 1, C.byref(st_)), "cgnr_step_status"), L.CgnrStatus, 32)
        res["cgnr"] = {"us_per_iterate_call_wall": us, "iterations_per_s": 1e6 / us}
        us = cadence(lambda: rls.createLinearSolver(rls.FISTA, Ad, AHA=G, reg=rls.L1Regularization(1e-2), rho=rho, iterations=32, relTol=0.0),
                     lambda p, st_: L.check(h, lib.rls_fista_step_status(p, 1, C.byref(st_)), "fista_step_status"), L.FistaStatus, 32)
        res["fista_l1"] = {"us_per_iterate_call_wall": us, "iterations_per_s": 1e6 / us}
        us = cadence(lambda: rls.createLinearSolver(rls.ADMM, Ad, AHA=G, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=8, iterationsCG=10,
                                                    tolInner=1e-5, absTol=0.0, relTol=0.0),
                     lambda p, st_: L.check(h, lib.rls_admm_step_status(p, 1, C.byref(st_), None, 0), "admm_step_status"), L.AdmmStatus, 8)
        res["admm_l1_outer"] = {"us_per_iterate_call_wall": us, "outer_iterations_per_s": 1e6 / us}
        return res

    for K in (8, 16, 64):
        @entry(f"cgnr_batched_{K}_rhs (BASELINE configs[3] on one GPU, f32 MFMA)")
        def _(K=K):
            X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
            Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
            S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
            rls.solve_(S, Bd, scheduler=rls.BatchedState)
            st = S.state
            us = timed(lambda: (rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 32), "init"),
                                rls._lib.check(h, lib.rls_cgnr_step(st._plan, 32), "step")), 32, reps=4)
            groups = 1 if K <= 8 else (K + 15) // 16  # passes over A per product: one per group of 16 (<= 8: the half layout)
            return {"us_per_batched_iteration": us, "solve_iterations_per_s": K * 1e6 / us,
                    "TFLOPs_algorithmic": 16.0 * M * N * K / us / 1e6, "frac_mfma_f32": 16.0 * M * N * K / us / 1e6 / MFMA_F32_PEAK_TF,
                    "A_stream_GBps (2 passes per group of 16, out of the Infinity Cache)": 2.0 * groups * M * N * 8 / us / 1e3,
                    "binding_roof": "hbm (8 flop/B at 8 right-hand sides, ridge ~20)" if K < 20 else "mfma"}

    for K in (8, 16, 64):
        @entry(f"cgnr_batched_{K}_rhs_gram_mode (BASELINE configs[3] on the reference's DEFAULT operator: AHA = A' * A explicit, shared by "
               f"all columns -- src/CGNR.jl:49,151, src/MultiThreading.jl:30-48)")
        def _(K=K):
            X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
            Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
            res = {}
            for tag, res_on in (("", 1), ("_streaming (resident = 0)", 0)) if K <= 8 else (("", 1),):
                ctx.tune(resident=res_on)
                try:
                    S = rls.createLinearSolver(rls.CGNR, Ad, AHA=state["G"], iterations=32, relTol=0.0)
                    rls.solve_(S, Bd, scheduler=rls.BatchedState)
                    st = S.state
                    pth = C.c_int32(-1)
                    lib.rls_cgnr_path(st._plan, C.byref(pth))
                    us = timed(lambda: (rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 32), "init"),
                                        rls._lib.check(h, lib.rls_cgnr_step(st._plan, 32), "step")), 32, reps=4)
                    us_long = None
                    if pth.value == 7:  # the resident launch amortises its load of AHA and the gather of x over the call: 128 iterations
                        us_long = timed(lambda: (rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 128), "init"),
                                                 rls._lib.check(h, lib.rls_cgnr_step(st._plan, 128), "step")), 128, reps=4)
                finally:
                    ctx.tune(resident=1)
                # bytes a batched iteration must move: AHA once (N^2 s), whatever the number of right-hand sides; flops 8 N^2 K
                gb = N * N * 8 / us / 1e3
                r = {"us_per_batched_iteration_incl_init": us, "solve_iterations_per_s": K * 1e6 / us, "kernel_path": pth.value,
                     "roofline": {"bound": "hbm" if K < 20 else "mfma",
                                  **({"achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                      "bytes_per_batched_iteration": N * N * 8} if K < 20 else
                                     {"achieved": 8.0 * N * N * K / us / 1e6, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                      "frac": 8.0 * N * N * K / us / 1e6 / MFMA_F32_PEAK_TF}),
                                  "note": ("path 7: AHA lives in the register files for the whole step call, so per iteration it moves only the "
                                           "all-gather of V = AHA P (N x K values) -- the HBM figure is the algorithm's bytes / time, not traffic"
                                           if pth.value == 7 else "path 6: one matrix-core product over AHA per batched iteration + the per-column update")}}
                if us_long is not None:
                    r["us_per_batched_iteration_128_iteration_call_incl_init"] = us_long
                res["gram" + tag] = r
            res["gram_setup_ms (A' * A on the matrix cores, once per operator)"] = state.get("gram_ms")
            return res

    @entry("fista_l1_batched_8_rhs_gram_mode (solve!(FISTA, B) on the reference's DEFAULT operator, AHA explicit and shared by the columns -- "
           "src/FISTA.jl:58,151, src/MultiThreading.jl:30-48)")
    def _():
        Xf = (rng.standard_normal((N, 8)) + 1j * rng.standard_normal((N, 8))).astype(np.complex64)
        Bf = rls.DeviceMatrix.from_host(np.asfortranarray((A @ Xf).astype(np.complex64)), ctx)
        res = {}
        for tag, res_on in (("", 1), ("_streaming (resident = 0)", 0)):
            ctx.tune(resident=res_on)
            try:
                S = rls.createLinearSolver(rls.FISTA, Ad, AHA=state["G"], reg=rls.L1Regularization(1e-2), rho=rho, iterations=32, relTol=0.0)
                rls.solve_(S, Bf, scheduler=rls.BatchedState)
                stf = S.state
                pth = C.c_int32(-1)
                lib.rls_fista_path(stf._plan, C.byref(pth))
                us = {n: timed(lambda n=n: (rls._lib.check(h, lib.rls_fista_init_batched(stf._plan, Bf.ptr, Bf.lda, rho, 1.0, 0.0, n, 0), "init"),
                                            rls._lib.check(h, lib.rls_fista_step(stf._plan, n), "step")), n, reps=4) for n in (32, 128)}
            finally:
                ctx.tune(resident=1)
            gb = N * N * 8 / us[32] / 1e3
            res["gram" + tag] = {"us_per_batched_iteration_incl_init": us[32], "us_per_batched_iteration_128_iteration_call_incl_init": us[128],
                                 "solve_iterations_per_s": 8e6 / us[32], "kernel_path": pth.value,
                                 "roofline": {"bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                              "bytes_per_batched_iteration": N * N * 8,
                                              "note": ("path 7: AHA in the register files for the whole step call; per iteration only the rows of the next "
                                                       "extrapolated point travel (N x K values, ONE grid barrier) -- bytes / time, not traffic"
                                                       if pth.value == 7 else "path 3: one matrix-core product over AHA + the per-column update launch")}}
        return res

    @entry("fista_l1_batched_16_rhs (solve!(FISTA, B), f32 MFMA)")
    def _():
        Xf = (rng.standard_normal((N, 16)) + 1j * rng.standard_normal((N, 16))).astype(np.complex64)
        Bf = rls.DeviceMatrix.from_host(np.asfortranarray((A @ Xf).astype(np.complex64)), ctx)
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
        rls.solve_(S, Bf, scheduler=rls.BatchedState)
        stf = S.state
        us = timed(lambda: (rls._lib.check(h, lib.rls_fista_init_batched(stf._plan, Bf.ptr, Bf.lda, rho, 1.0, 0.0, 48, 0), "init"),
                            rls._lib.check(h, lib.rls_fista_step(stf._plan, 48), "step")), 48, reps=4)
        return {"us_per_batched_iteration": us, "solve_iterations_per_s": 16 * 1e6 / us}

    @entry("admm_l1_batched_16_rhs (solve!(ADMM, B), shared A: cg! on the matrix cores, one workgroup per column for the rest)")
    def _():
        Xa = (rng.standard_normal((N, 16)) + 1j * rng.standard_normal((N, 16))).astype(np.complex64)
        Ba = rls.DeviceMatrix.from_host(np.asfortranarray((A @ Xa).astype(np.complex64)), ctx)
        S = rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=4, iterationsCG=10, tolInner=1e-5)
        rls.solve_(S, Ba, scheduler=rls.BatchedState)
        st = S.state
        assert type(st).__name__ == "AdmmBatchedState"
        us = timed(lambda: (st.init(Ba), st._step(4)), 4, reps=4)
        S1 = rls.createLinearSolver(rls.ADMM, Ad, reg=rls.L1Regularization(1e-2), rho=0.1, iterations=4, iterationsCG=10, tolInner=1e-5)
        b1 = Ba.column(0)
        rls.solve_(S1, b1)
        us1 = timed(lambda: rls.solve_(S1, b1), 4, reps=4)
        return {"us_per_batched_outer_iteration": us, "us_per_outer_iteration_one_column": us1, "outer_iterations_per_s_all_columns": 16e6 / us,
                "speedup_vs_column_by_column": 16 * us1 / us}

    @entry("cgnr_distinct_A_8_problems (BASELINE configs[3], distinct-A flavour on one GPU: 8 matrices, 8 streams)")
    def _():
        from rls_amd.multigpu import ConcurrentSolves
        mats = [A] + [make_A(M, N, 100 + k) for k in range(1, 8)]
        rhs = [(m_ @ np.ones(N, np.complex64)).astype(np.complex64) for m_ in mats]
        res = {}
        for ns in (1, 8):
            cs = ConcurrentSolves(rls, n_streams=ns)
            try:
                dA = cs.upload(mats)
                mk = lambda Ad_: rls.createLinearSolver(rls.CGNR, Ad_, iterations=32, relTol=0.0)
                cs.solve(dA, rhs, mk)
                t0 = time.perf_counter()
                for _ in range(3):
                    cs.solve(dA, rhs, mk)
                dt_ = (time.perf_counter() - t0) / 3
            finally:
                cs.close()
            res[f"{ns}_stream(s)"] = {"ms_per_8_solves": 1e3 * dt_, "solve_iterations_per_s": 8 * 32 / dt_}
        res["note"] = ("whole solve! calls from worker threads (plan creation, upload of b, 32 iterations, download of x); the slab / resident "
                       "kernels need the whole chip, so only setup and the small kernels overlap")
        # the same 8 problems (host b in, host x out) as ONE queue: rls_cgnr_solve_queue_host -- plans cached per operator, uploads,
        # init!, iterations and downloads of problem k enqueued behind problem k - 1's, one synchronisation for all eight
        cs = ConcurrentSolves(rls, n_streams=1)
        try:
            dA = cs.upload(mats)
            mk = lambda Ad_: rls.createLinearSolver(rls.CGNR, Ad_, iterations=32, relTol=0.0)
            xs = cs.solve_queue(dA, rhs, mk)
            errs_q = []
            for m_, b_, x_ in zip(mats, rhs, xs):
                errs_q.append(float(np.linalg.norm(x_.astype(np.complex128) - float64_cgnr(m_, b_, 32)) / np.linalg.norm(x_)))
            dts = []
            for _ in range(5):
                t0 = time.perf_counter()
                cs.solve_queue(dA, rhs, mk)
                dts.append(time.perf_counter() - t0)
            dt_ = min(dts)
        finally:
            cs.close()
        # the yardstick: 8 solves of 32 iterations back to back on ONE operator (init! included, device b, no copies, one synchronisation)
        S1 = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
        rls.init_(S1, b); lib.rls_cgnr_step(S1.state._plan, 32); ctx.sync()
        d1 = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(8):
                rls.init_(S1, b); lib.rls_cgnr_step(S1.state._plan, 32)
            ctx.sync(); d1.append(time.perf_counter() - t0)
        res["queue (rls_cgnr_solve_queue_host: one stream, one synchronisation)"] = {
            "ms_per_8_solves": 1e3 * dt_, "solve_iterations_per_s": 8 * 32 / dt_,
            "max_rel_err_vs_float64_cgnr_32_iterations": max(errs_q),
            "same_operator_8_solves_ms (init! + 32 iterations each, device b, no copies)": 1e3 * min(d1),
            "of_the_single_operator_rate": min(d1) / dt_}
        return res

    @entry("kaczmarz_row_sweeps (one launch per solve)")
    def _():
        S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(1e-3), iterations=4)
        rls.solve_(S, b); ctx.sync()
        t0 = time.perf_counter(); rls.solve_(S, b); ctx.sync(); dt = time.perf_counter() - t0
        return {"us_per_row_step": dt / (4 * M) * 1e6}

    @entry("admm_tv_config3 (BASELINE configs[2], device plan)")
    def _():
        # ADMM + TV, 8192 x 4096 Float32, shape (64, 64), 10 outer x 10 inner cg! iterations -- whole outer iterations
        # enqueued as a device plan (rls_admm_step); wall clock of complete solves, min of 3
        M3, N3 = 8192, 4096
        A3 = make_A(M3, N3, 3, np.float32)
        A3d = rls.DeviceMatrix.from_host(A3, ctx)
        b3 = rls.DeviceVector.from_host((A3 @ np.ones(N3, np.float32)).astype(np.float32), ctx)
        S = rls.createLinearSolver(rls.ADMM, A3d, reg=rls.TVRegularization(1e-2, shape=(64, 64)), rho=0.1, iterations=10,
                                   iterationsCG=10, tolInner=1e-5)
        rls.solve_(S, b3); rls.solve_(S, b3); ctx.sync()
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                rls.solve_(S, b3)
            ctx.sync(); dts.append((time.perf_counter() - t0) / 50)
        ms = 1e3 * min(dts)
        alg = 11 * 2 * M3 * N3 * 4  # 11 normal-operator applies per outer iteration, A read twice each on the reference path
        res = {"ms_per_outer_iteration": ms, "inner_cg_iterations": S.state.cg_iterations, "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
               "frac_algorithmic": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the same solver built the way the reference constructor builds it for a dense matrix (AHA = A'*A explicit,
        # src/ADMM.jl:82): cg! on the 64 MiB Gram matrix, resident in the register files
        t0 = time.perf_counter(); G3 = A3d.gram(); ctx.sync(); tg = time.perf_counter() - t0
        Sg = rls.createLinearSolver(rls.ADMM, A3d, AHA=G3, reg=rls.TVRegularization(1e-2, shape=(64, 64)), rho=0.1, iterations=10,
                                    iterationsCG=10, tolInner=1e-5)
        rls.solve_(Sg, b3); rls.solve_(Sg, b3); ctx.sync()
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                rls.solve_(Sg, b3)
            ctx.sync(); dts.append((time.perf_counter() - t0) / 50)
        res["gram_mode (AHA = A'*A explicit, the reference constructor's default for a dense matrix)"] = {
            "ms_per_outer_iteration": 1e3 * min(dts), "setup_gram_gemm_ms": 1e3 * tg, "inner_cg_iterations": Sg.state.cg_iterations}
        return res

    return out


def config5_leg(rls, ctx, dist, rank, world, barrier, K=64, W=32, rows=65536, rehearse=False, agree=None, inject=None):
    """the config-5 block of the default N > 1 line: a short `--workload rowsharded` run (it/s, the per-rank step_local_a time
    and HBM fraction, the all-reduce time, the backend and world size the collective saw) and the one-process host.
    agree(stage, err): the ranks' agreement point (main()); a failure on any rank raises LegFailed on all of them there."""
    from importlib import import_module

    mg = import_module("rls_amd.multigpu")
    agree = agree or (lambda stage, err=None: (_ for _ in ()).throw(err) if err is not None else None)
    full = mg.bench_rowsharded(rls, ctx, dist, rank, world, K, W, M=rows, agree=agree, inject=inject)
    dom = full["roofline"]["kernel"]
    out = {"metric": full["metric"], "value": full["value"], "unit": full["unit"], "ms_per_step": full["ms_per_step"], "scaling": "strong",
           "rows_per_gpu": full["config"]["rows_per_gpu"], "collective": full["config"]["collective"],
           "step_local_a_us": full["roofline"]["per_kernel"][dom]["us_per_call"],
           "step_local_a_frac_hbm": full["roofline"]["per_kernel"][dom]["frac_hbm"], "residual": full["residual"]}
    # the one-process host runs on rank 0 alone (no collectives inside); the others wait at the agreement point behind it
    err = None
    try:
        out["one_process_host"] = mg.bench_rowsharded_one_process(rls, rank, world, K, W, M=rows,
                                                                   devices=([0] * world if rehearse else None))
    except Exception as e:  # noqa: BLE001
        out["one_process_host"] = {"error": f"{type(e).__name__}: {e}"}
    agree("config5: one-process host done", err)
    return out


def load_pmc(kernel_prefix):
    """HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
    rocprofv3 --pmc runs; tools/pmc_summarize.py); None if not collected for this kernel"""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            for k, v in pmc["kernels"].items():
                if k.startswith(kernel_prefix):
                    return v["hbm_bytes_per_launch"], name
        except Exception:
            continue
    return None, None


class HostStagedDist:
    """`--rehearse`: the part of torch.distributed this script uses, over gloo, with device tensors staged through the host
    (`t.cpu()` waits for torch's current stream -- the stream the row-sharded kernels borrow -- and `t.copy_` is ordered on it)."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def all_reduce(self, t, op=None):
        import torch

        op = self._d.ReduceOp.SUM if op is None else op
        c = t.cpu() if t.is_cuda else t
        self._d.all_reduce(torch.view_as_real(c) if c.is_complex() else c, op=op)
        if t.is_cuda:
            t.copy_(c)

    def all_gather(self, outs, t):
        cs = [o.cpu() for o in outs]
        self._d.all_gather(cs, t.cpu())
        for o, c in zip(outs, cs):
            o.copy_(c)

    def barrier(self):
        self._d.barrier()

    def get_backend(self):
        return f"{self._d.get_backend()} (rehearsal: device tensors staged through the host)"

    def get_world_size(self):
        return self._d.get_world_size()

    def destroy_process_group(self):
        self._d.destroy_process_group()


def live_pmc(kernel_prefixes, timeout_s=90):
    """HBM bytes per launch of the named kernels measured IN THIS RUN -- two child processes, `rocprofv3 --pmc
    FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (separate passes, nothing but the counter: the guide's recipe) over
    tools/pmc_probe_headline.py, FETCH_SIZE doubled as the guide prescribes for gfx950's wide coalesced reads.  Returns
    {prefix: bytes} or {"error": ...}; never raises (the line then keeps the committed pass)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {}
    try:
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix=f"rls_pmc_{counter.lower()}_", dir="/tmp")
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_probe_headline.py")], cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"},
                               capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(files)} csv file(s): {r.stderr[-300:]}"}
            agg = {}
            for row in csv.DictReader(open(files[0])):
                agg.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            vals[counter] = {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}  # KB -> bytes, mean over the launches
            shutil.rmtree(d, ignore_errors=True)
        for pref in kernel_prefixes:
            f = max((v for k, v in vals["FETCH_SIZE"].items() if pref + "<" in k or pref + "(" in k), default=None)
            w = max((v for k, v in vals["WRITE_SIZE"].items() if pref + "<" in k or pref + "(" in k), default=0.0)
            if f is not None:
                out[pref] = 2.0 * f + w
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}
    return out or {"error": "the kernels did not appear in the counter output"}


def self_launch_command(gpus, argv, env):
    """the torch.distributed.run command line `python bench.py --gpus N` starts when it is NOT already a rank (no WORLD_SIZE
    / RANK in the environment) and N > 1; None otherwise.  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if gpus <= 1 or "WORLD_SIZE" in env or "RANK" in env:
        return None
    import socket

    with socket.socket() as so:  # a free port for the rendezvous
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    child_args = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + child_args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3200)
    ap.add_argument("--warmup", type=int, default=320)
    ap.add_argument("--workload", default="auto", choices=["auto", "cgnr", "single", "config4", "rowsharded"])
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--rhs-per-gpu", type=int, default=8)
    ap.add_argument("--config4-gram", action="store_true", help="--workload config4 on the explicit Gram matrix (the reference's default operator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--resident", type=int, default=1, help="0: force the two-launch pipeline for the headline run")
    ap.add_argument("--dry-launch", action="store_true", help="N > 1 from a plain shell: print the launch command as JSON and exit")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do NOT collect roofline.traffic in this run (two `rocprofv3 --pmc` child passes over a short probe of the "
                         "same solve, about six seconds each); quote the committed pass under profiles/ instead")
    ap.add_argument("--rehearse", action="store_true",
                    help="run the N > 1 code path on ONE GPU: every rank on device 0, torch.distributed over gloo (device tensors staged "
                         "through the host), the library's communicator on its direct transport with the ranks sharing the device; the "
                         "register-resident kernels are switched off (N processes cannot all own the chip).  The line says `rehearsal`: "
                         "its numbers are not a scaling measurement")
    ap.add_argument("--c5-rows", type=int, default=65536, help="total rows of the config-5 matrix (rehearsals shrink it)")
    args = ap.parse_args()
    # (the pool's host driver supports dmabuf IPC only: without this RCCL's and our communicator's cross-process buffers fail with
    #  hipIpcGetMemHandle: invalid argument.  The image exports it already; a launcher that builds its own environment may not.)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    launch = self_launch_command(args.gpus, sys.argv[1:], os.environ)
    if launch is not None:
        # The parent of the N ranks.  Nothing here may import torch or touch HIP: the ranks are child processes (a process
        # that has initialised the GPU must not replace itself with another program on this pool).
        if args.dry_launch:
            print(json.dumps({"launch": launch, "torch_imported": "torch" in sys.modules}))
            return 0
        import subprocess

        return subprocess.run(launch).returncode  # rank 0's JSON line goes straight to our stdout
    if args.dry_launch:
        print(json.dumps({"launch": None, "torch_imported": "torch" in sys.modules}))
        return 0

    import torch  # plumbing: device selection, barrier, max-over-ranks

    t_start = time.perf_counter()

    def trace(what):
        """RLS_BENCH_TRACE=1: stage marks on stderr (rank, seconds since start) -- where a multi-rank run spends its time"""
        if os.environ.get("RLS_BENCH_TRACE"):
            print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - t_start:7.2f}s] {what}", file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: start with `python bench.py --gpus N` (it launches the ranks) "
                         "or under torch.distributed.run with --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    device = 0 if args.rehearse else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist

        from datetime import timedelta

        # a rank that dies inside a collective must not hold the others for torch's default 10 minutes (RLS_BENCH_DIST_TIMEOUT_S:
        # the tests shorten it)
        pg_timeout = timedelta(seconds=float(os.environ.get("RLS_BENCH_DIST_TIMEOUT_S", "120")))
        if args.rehearse:
            dist.init_process_group("gloo", timeout=pg_timeout)
            dist = HostStagedDist(dist)
        else:
            dist.init_process_group("nccl", timeout=pg_timeout, device_id=torch.device("cuda", local_rank))

    trace("process group up")
    import rls_amd as rls

    ctx = rls.Context(device)
    if args.rehearse:
        ctx.tune(resident=0)
    M, N = args.M, args.N
    dt = np.complex64
    s = np.dtype(dt).itemsize
    K, W = args.steps, args.warmup
    workload = args.workload
    if workload in ("auto", "single"):
        workload = "cgnr"

    legs = {"broken": None}  # set once a collective has failed or the ranks are known to be out of step: no collective after that

    def finish(result=None):
        if rank == 0 and result is not None:
            if legs["broken"]:
                result["distributed_error"] = legs["broken"]
            print(json.dumps(result), flush=True)
        if dist is not None:
            if legs["broken"]:
                # the process group is not usable any more (a barrier here would wait for its timeout, destroy may hang): the line is
                # out, leave without the interpreter's teardown.  A rank that failed exits non-zero; the launcher relays it.
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(0 if rank == 0 else 1)
            dist.barrier()
            dist.destroy_process_group()

    class LegFailed(RuntimeError):
        pass

    def agree(stage, err=None):
        """every rank calls this at the same point of a leg with its own error (or None): the ranks all-reduce a flag (MAX of
        failing rank + 1), so that EITHER all go on OR all raise LegFailed and skip the rest of the leg together -- nobody is left
        alone in the leg's next collective.  The error text stays with its owner (stderr); the line names the owner's rank."""
        if err is not None:
            import traceback

            print(f"[bench rank {rank}] {stage}: {type(err).__name__}: {err}", file=sys.stderr, flush=True)
            traceback.print_exception(type(err), err, err.__traceback__, file=sys.stderr)
        flag = (rank + 1) if err is not None else 0
        if dist is not None and not legs["broken"]:
            try:
                tt = torch.tensor([float(flag)], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                flag = int(tt.item())
            except Exception as e:  # noqa: BLE001 -- the collective itself failed (a rank is gone, a timeout): stop using the group
                legs["broken"] = f"{stage}: the agreement all-reduce failed on rank {rank}: {type(e).__name__}: {e}"
                raise LegFailed(legs["broken"]) from e
        if flag:
            mine = f": {type(err).__name__}: {err}" if err is not None and flag == rank + 1 else ""
            raise LegFailed(f"{stage}: failed on rank {flag - 1}{mine}")

    def inject(leg):
        """RLS_BENCH_FAIL=<leg>:<rank> (tests): raise inside that leg's setup on that rank"""
        spec = os.environ.get("RLS_BENCH_FAIL", "")
        if spec and spec.split(":")[0] == leg and int(spec.split(":")[1]) == rank:
            raise RuntimeError(f"injected failure in leg {leg} on rank {rank} (RLS_BENCH_FAIL)")

    if workload == "rowsharded":
        from importlib import import_module

        mg = import_module("rls_amd.multigpu")
        return finish(mg.bench_rowsharded(rls, ctx, dist, rank, world, K, W, M=args.c5_rows))

    lib, h = ctx.lib, ctx.handle

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    def timed_regions(prepare, run, est_s):
        """the contract's timed region -- barrier + synchronize, EXACTLY K steps, synchronize + barrier -- repeated
        when it is short.  Returns wall-clock seconds (max over ranks) and hipEvent seconds of every repetition."""
        if dist is not None:
            # every rank must run the SAME number of repetitions (each one holds a barrier and an all-reduce): the estimate is a
            # per-rank measurement, so agree on the largest (found by the one-GPU rehearsal: rank 0 waited for ever)
            tt = torch.tensor([est_s], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            est_s = float(tt.item())
        reps = 5 if est_s >= 0.010 else int(min(400, max(50, math.ceil(0.4 / max(est_s, 1e-5)))))
        walls, evs = [], []
        for _ in range(reps):
            prepare()
            barrier()
            ctx.timer_start()
            t0 = time.perf_counter()
            run()
            ev_ms = ctx.timer_stop_ms()  # hipEvents on the stream the kernels run on; synchronises that stream
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if dist is not None:
                dist.barrier()
                tt = torch.tensor([el], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            walls.append(el)
            evs.append(ev_ms * 1e-3)
        return walls, evs

    def solo_rate(prepare, run, units):
        """units / wall time of `run` on rank 0 ALONE (the other ranks wait at the barrier): the N = 1 figure of the same
        workload, measured in the same job; broadcast to every rank.  None at N = 1 (the line's value is that figure)."""
        if dist is None:
            return None
        barrier()
        rate = 0.0
        if rank == 0:
            els, t_all = [], time.perf_counter()
            while len(els) < 5 or (len(els) < 200 and time.perf_counter() - t_all < 0.3):
                prepare()
                ctx.sync()
                t0 = time.perf_counter()
                run()
                ctx.sync()
                els.append(time.perf_counter() - t0)
            rate = units / statistics.median(els)
        tt = torch.tensor([rate], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        barrier()
        return float(tt.item())

    def config4_measure(K, W, gram=False, leg_name=None):
        # ---- BASELINE configs[3]: shared A (seed 4), B = A X (X: seed 5, 64 columns), columns 8k..8k+7 on GPU k ----
        # gram: the operator the reference constructor builds for a dense matrix (AHA = A' * A explicit, src/CGNR.jl:49), shared
        # by the columns (src/MultiThreading.jl:30-48); the Gram GEMM is setup, outside the timed region, and reported
        # leg_name: this is a leg behind the headline of an N > 1 line -- its setup (no collectives) ends at an agreement point
        R = args.rhs_per_gpu
        setup_ms = None
        err = None
        try:
            if leg_name:
                inject(leg_name)
            A = make_A(M, N, seed=4)
            rng = np.random.default_rng(5)
            X = ((rng.standard_normal((N, 64)) + 1j * rng.standard_normal((N, 64))) / math.sqrt(2)).astype(dt)
            cols = [(rank * R + j) % 64 for j in range(R)]
            B = np.asfortranarray((A @ X[:, cols]).astype(dt))
            Ad = rls.DeviceMatrix.from_host(A, ctx)
            Bd = rls.DeviceMatrix.from_host(B, ctx)
            if gram:
                Ad.gram(); ctx.sync()
                t0 = time.perf_counter(); Gd = Ad.gram(); ctx.sync(); setup_ms = 1e3 * (time.perf_counter() - t0)
                solver = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, iterations=SEGMENT, relTol=0.0)
            else:
                solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=SEGMENT, relTol=0.0)
            rls.solve_(solver, Bd, scheduler=rls.BatchedState)  # builds the batched plan (and checks it runs)
            st = solver.state
            assert isinstance(st, rls.BatchedState), "config 4 needs the shared-A batched plan"
        except Exception as e:  # noqa: BLE001
            if not leg_name:
                raise
            err = e
        if leg_name:
            agree(f"{leg_name}: setup", err)

        def init():
            rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, SEGMENT), "init_batched")

        def step(n, initialised=False):
            while n > 0:
                m = min(n, SEGMENT)
                if not initialised:
                    init()
                initialised = False
                rls._lib.check(h, lib.rls_cgnr_step(st._plan, m), "rls_cgnr_step")
                n -= m

        step(20 * SEGMENT); ctx.sync()
        step(W)
        t0 = time.perf_counter(); init(); step(min(K, 4 * SEGMENT), True); ctx.sync()
        est = (time.perf_counter() - t0) * K / min(K, 4 * SEGMENT)
        walls, evs = timed_regions(init, lambda: step(K, True), est)
        elapsed, ev = statistics.median(walls), statistics.median(evs)
        stat = st.status()
        assert all(math.isfinite(s_.residual) for s_ in stat), "batched CGNR residual is not finite"
        rate = R * K / elapsed
        rates = [rate]
        if dist is not None:
            tt = torch.tensor([R * K / ev], dtype=torch.float64, device="cuda")
            gathered = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(gathered, tt)
            rates = [float(g.item()) for g in gathered]
        flops_iter = 16.0 * M * N * R  # complex: 8 real flops per MAC, two products
        us_iter = 1e6 * ev / K
        out = {
            "metric": "CGNR solve-iterations/sec, 64 independent solves of 4096x2048 CF32 sharded 8 per GPU (BASELINE configs[3])",
            "value": world * R * K / elapsed, "unit": "solve-iterations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "c64 (ComplexF32 storage, f32 MFMA arithmetic; scalar reductions accumulated in f64)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: CGNR {M}x{N} ComplexF32, shared A, {R} right-hand sides per GPU advancing together "
                                   f"(BatchedState: per-column scalars and done flags), lambda=0, relTol=0, back-to-back solves of {SEGMENT} "
                                   f"iterations, no data-path collective", "M": M, "N": N, "rhs_per_gpu": R, "step": "one batched iteration"},
            "timed_region": {"repetitions": len(walls), "wall_s": spread(walls), "hip_events_s": spread(evs),
                             "value_is": "n_gpus * rhs_per_gpu * steps / median(wall, max over ranks)"},
            "per_rank_solve_iterations_per_s_hip_events": rates,
        }
        # which roof binds a batched iteration: the two passes over the shared A (2*M*N*s bytes, whatever the
        # number of right-hand sides) against 16*M*N*R flops on the f32 matrix cores -- 8 right-hand sides are 8 flop/B,
        # under the ~20 flop/B ridge, i.e. bandwidth-bound; from ~20 right-hand sides on the MFMA rate binds
        bytes_iter = 2.0 * M * N * 8
        kern = "skinny_t_kernel + skinny_v_kernel (T = A P, V = A^H T on v_mfma_f32_16x16x4_f32) + per-column update"
        if gram:
            pth = C.c_int32(-1)
            lib.rls_cgnr_path(st._plan, C.byref(pth))
            flops_iter = 8.0 * N * N * R   # one product over the N x N matrix
            bytes_iter = 1.0 * N * N * 8   # AHA once per batched iteration (the streaming form reads it; the resident form holds it
                                           # in registers and moves only the N x R panel -- priced on the same basis)
            kern = ("cgnr_gramk_resident_kernel (AHA in registers, one launch per step call)" if pth.value == 7 else
                    "skinny_t_kernel<VOUT> (V = AHA P, one product) + per-column update")
            out["config"]["operator"] = "AHA = A' * A explicit (the reference constructor's default), shared by the columns"
            out["config"]["path"] = pth.value
            out["setup_gram_gemm_ms"] = setup_ms
        t_hbm, t_mfma = bytes_iter / (HBM_PEAK_GBS * 1e9), flops_iter / (MFMA_F32_PEAK_TF * 1e12)
        mf = {"achieved": flops_iter / us_iter / 1e6, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": flops_iter / us_iter / 1e6 / MFMA_F32_PEAK_TF}
        hb = {"achieved": bytes_iter / us_iter / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_iter / us_iter / 1e3 / HBM_PEAK_GBS}
        out["roofline"] = dict(bound="hbm" if t_hbm >= t_mfma else "mfma", kernel=kern, **(hb if t_hbm >= t_mfma else mf), traffic=None,
                               us_per_batched_iteration=us_iter, other_roof=(mf if t_hbm >= t_mfma else hb),
                               note=("hbm basis: AHA once per batched iteration (N*N*s bytes); mfma basis: 8*N*N flops per solve-iteration" if gram else
                                     "hbm basis: A streamed twice per batched iteration (2*M*N*s bytes, shared by the right-hand sides; it "
                                     "comes out of the Infinity Cache, so the memory-side counters see less); mfma basis: 16*M*N flops per "
                                     "solve-iteration (complex MAC = 8 flops, two products), padding columns not counted"))
        n1 = solo_rate(init, lambda: step(K, True), R * K)
        if n1 is not None:
            out["n1_same_workload_value"] = n1
            out["efficiency_vs_n1_same_workload"] = out["value"] / (world * n1)
        return out

    if workload == "config4":
        return finish(config4_measure(K, W, gram=args.config4_gram))

    # ---- headline: one CGNR solve per GPU, data resident in HBM before the timed region ---------------------------
    A = make_A(M, N, seed=2 if world == 1 else 100 + rank)
    rng = np.random.default_rng(1000 + rank)
    x_true = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(dt)
    b = (A @ x_true).astype(dt)
    Ad = rls.DeviceMatrix.from_host(A, ctx)
    bd = rls.DeviceVector.from_host(b, ctx)
    if not args.resident:
        ctx.tune(resident=0)
    solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=SEGMENT, relTol=0.0)
    rls.init_(solver, bd)
    st = solver.state
    path = C.c_int32(-1)
    lib.rls_cgnr_path(st._plan, C.byref(path))

    def step(n, initialised=False):
        """n CGNR iterations as back-to-back solves of SEGMENT iterations; `initialised`: the first solve's init! has
        already run (it belongs to the solve's setup: SURVEY 8d times iterations of a running solve)"""
        while n > 0:
            m = min(n, SEGMENT)
            if not initialised:
                rls.init_(solver, bd)  # asynchronous: r = A^H b, then the init kernel
            initialised = False
            rls._lib.check(h, lib.rls_cgnr_step(st._plan, m), "rls_cgnr_step")
            n -= m

    trace("headline: operator uploaded, solver initialised")
    # setup (untimed, not part of W): the first LONG host wait of a process returns ~50 ms late, once
    # (tools/stall_probe2.py); take that hit here, outside the measurement.
    step(150 * SEGMENT)
    ctx.sync()
    step(W)
    ctx.sync()
    t0 = time.perf_counter(); rls.init_(solver, bd); step(min(K, 4 * SEGMENT), True); ctx.sync()
    est = (time.perf_counter() - t0) * K / min(K, 4 * SEGMENT)
    trace("headline: warm-up done")
    walls, evs = timed_regions(lambda: rls.init_(solver, bd), lambda: step(K, True), est)
    trace("headline: timed regions done")
    elapsed, ev = statistics.median(walls), statistics.median(evs)
    st._refresh(lib)
    assert st.iteration == ((K - 1) % SEGMENT) + 1, (st.iteration, K)
    assert math.isfinite(st._residual), "CGNR residual is not finite"
    # the line proves its own work: the x the timed region left behind (st.iteration iterations into its last solve) against a
    # float64 CGNR of the same count on the host, and against the planted solution (CG converges geometrically on this matrix)
    sol_iter = int(st.iteration)
    x_dev = st.x.to_host().astype(np.complex128)
    x_f64 = float64_cgnr(A, b, sol_iter)
    sol_err = float(np.linalg.norm(x_dev - x_f64) / np.linalg.norm(x_f64))
    sol_err_true = float(np.linalg.norm(x_dev - x_true) / np.linalg.norm(x_true))
    assert sol_err <= 1e-5, f"the timed kernel's solution is off the float64 CGNR iterate by {sol_err:.3e} (> 1e-5)"
    if dist is not None:
        tt = torch.tensor([sol_err], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sol_err = float(tt.item())

    # ---- the dominant kernel, timed live with hipEvents on the stream it is launched on ---------------------------
    bytes_iter = bytes_per_cgnr_iteration(M, N, s)
    iter_gbs = bytes_iter * K / ev / 1e9
    nwg = M // 16
    kern = {}
    if path.value == 4:
        # one launch = a whole step call (here SEGMENT iterations); the memset node that zeroes its arrival counters
        # (1 KiB) sits inside the event pair
        rls.init_(solver, bd); lib.rls_cgnr_step(st._plan, SEGMENT); ctx.sync()
        us = []
        for _ in range(20):
            rls.init_(solver, bd)
            ctx.timer_start()
            lib.rls_cgnr_step(st._plan, SEGMENT)
            us.append(1e3 * ctx.timer_stop_ms())
        us_launch = statistics.median(us)
        # two-level exchange: partial rows out + in (slices over the group), 8 group partials out, all of them read by every workgroup
        exch = (2 * nwg * N * s + 8 * N * s + nwg * 8 * N * s) * SEGMENT
        dom = "cgnr_resident_kernel"
        kern[dom] = {"us_per_launch": us_launch, "iterations_per_launch": SEGMENT, "us_per_iteration_in_kernel": us_launch / SEGMENT,
                     "algorithmic_bytes_per_launch": bytes_iter * SEGMENT,
                     "effective_GBps": bytes_iter * SEGMENT / us_launch / 1e3,
                     "min_hbm_bytes_per_launch": M * N * s, "exchange_bytes_per_launch (L2 / Infinity Cache, not HBM-bound)": exch}
    elif path.value == 1:
        us_a, us_r = C.c_float(), C.c_float()
        rls.init_(solver, bd)
        rls._lib.check(h, lib.rls_cgnr_step_profiled(st._plan, 200, C.byref(us_a), C.byref(us_r)), "profiled")
        dom = "cgnr_pipe_a_kernel"
        kern[dom] = {"us_per_launch": us_a.value, "iterations_per_launch": 1, "algorithmic_bytes_per_launch": 2 * M * N * s,
                     "effective_GBps": 2 * M * N * s / us_a.value / 1e3, "min_hbm_bytes_per_launch": M * N * s + nwg * N * s}
        kern["cgnr_pipe_r_kernel"] = {"us_per_launch_back_to_back": us_r.value, "note": "in situ ~4.8 us (rocprofv3): it reads partial rows written "
                                                                                        "on other XCDs"}
    else:
        dom = PATH_NAMES.get(path.value, "?")
        kern[dom] = {"us_per_launch": 1e6 * ev / K, "iterations_per_launch": 1, "algorithmic_bytes_per_launch": bytes_iter,
                     "effective_GBps": iter_gbs, "min_hbm_bytes_per_launch": 2 * M * N * s}
    floor = None
    if path.value == 4 and rank == 0:
        # The roof of the resident kernel (its HBM fraction says nothing: A never moves).  Per iteration it must at least
        # (a) run its two products on the VALU: 16 M N flops at the guide's FP32 vector peak (157.3 TF, packed FMAs included), and
        # (b) all-reduce the 256 partial rows of v.  Two figures for (b):
        #   * protocol-independent: any all-reduce over 256 workgroups in 8 XCDs needs at least one synchronisation inside
        #     the XCD-sized groups and one across the grid -- the bare group and grid barriers of tools/ubench/grid_barrier --
        #     plus its payload at the guide's rates: every partial row leaves its XCD and is read once (Infinity Cache rate),
        #     and every workgroup reads the N-vector result (an XCD's L2, shared rows);
        #   * this design's: the two-level exchange with the arithmetic stripped, measured live by the same micro-benchmark
        #     (same protocol, same shapes, nothing else in the kernel) -- it bounds the protocol, not the machine.
        ge = grid_exchange_floor()
        if ge and "two_level" in ge:
            valu_us = 16.0 * M * N / (VALU_F32_PEAK_TF * 1e12) * 1e6
            # the exchange the kernel runs since round 4 stores its partial rows at L2 scope (an XCD-aligned group shares its L2):
            # its stripped form is the floor; the write-through form of rounds 2-3 is kept beside it
            xch_us = ge.get("two_level_l2_rows", ge["two_level"])
            floor_us = valu_us + xch_us
            it_us = kern[dom]["us_per_iteration_in_kernel"]
            row_bytes = N * s
            payload_us = (2.0 * nwg * row_bytes / (INF_CACHE_TBS * 1e12) + nwg * row_bytes / (L2_SHARED_TBS * 1e12)) * 1e6
            indep = None
            if "bar" in ge and "gbar" in ge:
                indep = valu_us + ge["gbar"] + ge["bar"] + payload_us
            floor = {"bound": "grid-exchange", "unit": "us per iteration", "floor": floor_us, "measured": it_us, "frac_of_floor": floor_us / it_us,
                     "components": {"products_on_the_VALU_at_peak": valu_us, "valu_peak_TFLOPs": VALU_F32_PEAK_TF,
                                    "all_reduce_two_level_arithmetic_stripped": xch_us,
                                    "all_reduce_two_level_rows_written_through (rounds 2-3)": ge["two_level"],
                                    "for_comparison": {"bare_grid_barrier_256_workgroups": ge.get("bar"),
                                                       "bare_group_barrier_8_groups_of_32": ge.get("gbar"),
                                                       "all_reduce_flat_two_hops (round 2's exchange)": ge.get("flat")}},
                     "protocol_independent_floor": (None if indep is None else {
                         "us_per_iteration": indep, "frac": indep / it_us,
                         "components": {"products_on_the_VALU_at_peak": valu_us, "bare_group_barrier": ge["gbar"], "bare_grid_barrier": ge["bar"],
                                        "payload_at_guide_rates": payload_us,
                                        "payload_bytes": {"partial_rows_out_and_in (Infinity Cache rate)": 2 * nwg * row_bytes,
                                                          "result_read_by_every_workgroup (L2 rate)": nwg * row_bytes},
                                        "rates_TBps": {"infinity_cache": INF_CACHE_TBS, "l2_shared_rows": L2_SHARED_TBS}}}),
                     "source": "tools/ubench/grid_barrier (run live by bench.py); profiles/ holds a committed run"}
    hbm_streaming = None
    if rank == 0 and world == 1 and path.value == 4:
        # the HBM story beside the headline's grid-exchange roof: the SAME solve on the path that does stream A every iteration
        # (resident = 0: cgnr_pipe_a_kernel reads A once and forms both products, cgnr_pipe_r_kernel sums the partial rows)
        try:
            ctx.tune(resident=0)
            try:
                Sp = rls.createLinearSolver(rls.CGNR, Ad, iterations=SEGMENT, relTol=0.0)
                rls.solve_(Sp, bd)
                best = float("inf")
                for _ in range(8):
                    rls.init_(Sp, bd)
                    ctx.timer_start()
                    lib.rls_cgnr_step(Sp.state._plan, SEGMENT)
                    best = min(best, ctx.timer_stop_ms())
                us_it = best * 1e3 / SEGMENT
                us_a, us_r = C.c_float(), C.c_float()
                rls.init_(Sp, bd)
                rc = lib.rls_cgnr_step_profiled(Sp.state._plan, 100, C.byref(us_a), C.byref(us_r))
            finally:
                ctx.tune(resident=1)
            real_a = M * N * s + nwg * N * s
            tr_a, tr_src = load_pmc("cgnr_pipe_a_kernel")
            hbm_streaming = {"path": "two-launch pipeline (resident = 0): A streamed ONCE per iteration, both products from the register slab",
                             "us_per_iteration": us_it, "iterations_per_s": 1e6 / us_it,
                             "frac_algorithmic": bytes_iter / us_it / 1e3 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_iteration (SURVEY 8d: A read twice)": bytes_iter}
            if rc == 0:
                hbm_streaming.update({"kernel": "cgnr_pipe_a_kernel", "kernel_us_back_to_back": us_a.value,
                                      "real_bytes_per_launch (A once + the partial rows)": real_a,
                                      "frac_real_bytes": real_a / (us_a.value * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                      "real_GBps": real_a / us_a.value / 1e3, "traffic_pmc_static": tr_a, "traffic_source": tr_src,
                                      "reduce_kernel_us_back_to_back": us_r.value})
        except Exception as e:  # noqa: BLE001 -- beside the headline, must not cost the line
            hbm_streaming = {"error": repr(e)}
    trace("headline: solution checked, kernel timed")
    n1_value = solo_rate(lambda: rls.init_(solver, bd), lambda: step(K, True), K)
    trace("headline: solo rate done")
    # Every leg behind the headline is a unit the ranks enter and leave TOGETHER: its setup (no collectives inside) ends in agree(),
    # its measurement (collectives inside) ends in agree(); a leg that fails on one rank is skipped by all of them and the line carries
    # `error` with the owner's rank.  Once a collective itself has failed (legs["broken"]) nothing distributed runs any more.
    def leg(name, fn):
        if legs["broken"]:
            return {"error": "skipped: " + legs["broken"]}
        try:
            return fn()
        except LegFailed as e:
            return {"error": str(e)}
        except Exception as e:  # noqa: BLE001 -- raised on this rank outside the leg's agree() points: the others may be in a collective
            legs["broken"] = f"leg {name}: {type(e).__name__}: {e} on rank {rank} outside an agreement point"
            print(f"[bench rank {rank}] {legs['broken']}", file=sys.stderr, flush=True)
            return {"error": legs["broken"]}

    c4 = None
    if world > 1:
        # BASELINE configs[3], shared-A flavour, on the same job: 8 right-hand sides per GPU advancing together (matrix cores)
        def c4_leg():
            c4_full = config4_measure(max(SEGMENT, min(K, 20 * SEGMENT)), SEGMENT, leg_name="config4")
            out4 = {k: c4_full[k] for k in ("metric", "value", "unit", "ms_per_step", "n1_same_workload_value", "efficiency_vs_n1_same_workload",
                                            "per_rank_solve_iterations_per_s_hip_events") if k in c4_full}
            out4["roofline"] = {k: c4_full["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "us_per_batched_iteration")}
            return out4

        def g4_leg():  # the same job on the reference constructor's default operator (explicit AHA)
            g4 = config4_measure(max(SEGMENT, min(K, 20 * SEGMENT)), SEGMENT, gram=True, leg_name="config4_gram")
            o = {k: g4[k] for k in ("value", "unit", "ms_per_step", "n1_same_workload_value", "efficiency_vs_n1_same_workload",
                                    "per_rank_solve_iterations_per_s_hip_events", "setup_gram_gemm_ms") if k in g4}
            o["operator"] = g4["config"]["operator"]
            o["path"] = g4["config"]["path"]
            o["roofline"] = {k: g4["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "us_per_batched_iteration")}
            return o

        c4 = leg("config4", c4_leg)
        c4["gram_mode"] = leg("config4_gram", g4_leg)
    trace("config 4 legs done")
    c5 = None
    if world > 1:
        # BASELINE configs[4] on the same job: the 65536 x 8192 problem row-partitioned over the ranks, one all-reduce of
        # A^H t per iteration through torch.distributed (RCCL); then the one-process host of the same problem -- rank 0 alone
        # driving every GPU through the library's own communicator (the Julia host's call sequence) -- while the others wait
        c5 = leg("config5", lambda: config5_leg(rls, ctx, dist, rank, world, barrier, rows=args.c5_rows, rehearse=args.rehearse,
                                                agree=agree, inject=inject))
    trace("config 5 legs done")
    traffic, traffic_src = load_pmc(dom)
    traffic_kind = TRAFFIC_KIND
    if not args.no_live_pmc and rank == 0 and world == 1:
        trace("live PMC passes")
        lp = live_pmc([dom, "cgnr_pipe_a_kernel"])
        if dom in lp:
            traffic, traffic_src = lp[dom], "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes (tools/pmc_probe_headline.py)"
            traffic_kind = "live: collected by child rocprofv3 passes of this bench run (FETCH_SIZE x 2 + WRITE_SIZE, mean over the probe's launches)"
            if hbm_streaming and "cgnr_pipe_a_kernel" in lp and "error" not in hbm_streaming:
                hbm_streaming["traffic_pmc_live"] = lp["cgnr_pipe_a_kernel"]
        else:
            traffic_kind = TRAFFIC_KIND + f"  [the live passes failed: {lp.get('error')}]"
    kd = kern[dom]
    hbm_bytes = traffic if traffic is not None else kd["min_hbm_bytes_per_launch"]
    frac_hbm = hbm_bytes / (kd["us_per_launch"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    frac_alg = iter_gbs / HBM_PEAK_GBS

    if rank == 0:
        out = {
            "metric": "CGNR iterations/sec @4096x2048 CF32; achieved HBM GB/s vs roofline",
            "value": world * K / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c64 (ComplexF32 storage and arithmetic; scalar reductions accumulated in f64)",
            "data": "synthetic",
            **({"rehearsal": "every rank on device 0, gloo control plane, resident kernels off: the N > 1 CODE PATH on one GPU -- not a "
                             "scaling measurement"} if args.rehearse else {}),
            "config": {"workload": f"CGNR matrix-free normal operator, dense column-major ComplexF32 {M}x{N}, lambda=0, relTol=0, back-to-back "
                                   f"solves of {SEGMENT} iterations (BASELINE configs[1] shape, headline metric run of SURVEY 8d)" +
                                   ("" if world == 1 else "; one independent solve per GPU, no collectives"),
                       "M": M, "N": N, "problems_per_gpu": 1, "kernel_path": PATH_NAMES.get(path.value, str(path.value))},
            "timed_region": {"repetitions": len(walls), "wall_s": spread(walls), "hip_events_s": spread(evs),
                             "value_is": "n_gpus * steps / median(wall); every repetition is exactly `steps` iterations between "
                                         "barrier + synchronize on both sides",
                             "iterations_per_s_hip_events": K / ev},
            "solution_check": {"rel_err_vs_float64_cgnr_same_iteration": sol_err, "tolerance": 1e-5, "iteration": sol_iter,
                               "rel_err_vs_planted_x_true": sol_err_true,
                               "what": "x left by the LAST solve of the timed region (max over ranks) against a complex128 CGNR of the same "
                                       "iteration count run on the host by bench.py itself; asserted before the line is printed"},
            "roofline": {},
        }
        hbm_view = {
            # algorithmic basis, the ITERATION as the unit (SURVEY 8d bytes per iteration x iterations / device time of the timed region)
            "peak_GBps": HBM_PEAK_GBS, "algorithmic_GBps": iter_gbs, "frac_algorithmic": frac_alg,
            # physical basis: bytes the dominant kernel really moves per launch / its launch time
            "frac_hbm": frac_hbm, "traffic": traffic, "traffic_source": traffic_src, "traffic_kind": traffic_kind,
            "hbm_bytes_per_launch_used": hbm_bytes,
            "note": "frac_algorithmic: SURVEY 8d algorithmic bytes per iteration (A read twice) x iterations / hipEvent time of the timed region / peak; "
                    "frac_hbm: HBM bytes of the dominant kernel per launch (PMC when collected, else the minimum it must move) / its launch time / peak"}
        if floor is not None:
            # the resident kernel keeps A in the register files: the algorithm's bytes are not moved (frac_algorithmic > 1), so the HBM
            # roof does not bound it and is reported beside the roof that does -- its in-kernel grid-wide all-reduce.  `frac` is a
            # division: floor / measured, both in us per iteration
            out["roofline"] = {"bound": "grid-exchange", "kernel": dom, "unit": "iterations/s inside the dominant kernel",
                               "achieved": 1e6 / floor["measured"], "peak": 1e6 / floor["floor"],
                               "frac": (1e6 / floor["measured"]) / (1e6 / floor["floor"]),
                               "us_per_iteration": {"measured": floor["measured"], "floor": floor["floor"]},
                               "traffic": traffic, "traffic_source": traffic_src, "traffic_kind": traffic_kind,
                               "hbm_streaming": hbm_streaming,
                               "protocol_independent": floor["protocol_independent_floor"],
                               "hbm": dict(hbm_view, hbm_algorithmic_uncapped=iter_gbs,
                                           note=hbm_view["note"] + ".  A lives in VGPRs for the whole launch: per iteration only the partial-row "
                                                "exchange moves (L2 / Infinity Cache); the two-launch pipeline that streams A every iteration is under other_paths"),
                               "resident_floor": floor}
        else:
            out["roofline"] = {"bound": "hbm", "kernel": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": iter_gbs, "frac": frac_alg,
                               "frac_hbm": frac_hbm, "traffic": traffic, "traffic_source": traffic_src, "traffic_kind": traffic_kind,
                               "hbm_streaming": hbm_streaming, "hbm_bytes_per_launch_used": hbm_bytes, "note": hbm_view["note"]}
        out["roofline"]["per_kernel"] = kern
        out["roofline"]["iteration"] = {"bytes": bytes_iter, "us_hip_events": 1e6 * ev / K, "roofline_us_at_peak": bytes_iter / HBM_PEAK_GBS / 1e3}
        # which regime `value` is in: a resident launch pays its slab load once per step call, and the first solve's init! lies
        # outside the timed region when the region is a single launch
        out["config"]["iterations_per_launch"] = (min(K, SEGMENT) if path.value in (4, 5) else 1)
        out["config"]["init_inside_region"] = bool(K > SEGMENT)
        if n1_value is not None:
            out["n1_same_workload_value"] = n1_value
            out["efficiency_vs_n1_same_workload"] = out["value"] / (world * n1_value)
        if c4 is not None:
            out["config4_batched"] = c4
        if c5 is not None:
            out["config5_rowsharded"] = c5
        errors = []
        if world == 1 and not args.no_extras and (M, N) == (4096, 2048):
            out["other_paths"] = other_paths(rls, ctx, Ad, A, bd, errors)
        if errors:
            out["other_paths_error"] = errors
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baselines(A, b)
        finish(out)
    else:
        finish()


if __name__ == "__main__":
    sys.exit(main() or 0)
