#!/usr/bin/env python3
"""Headline benchmark: CGNR iterations/sec @ 4096x2048 ComplexF32 (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

A step is ONE CGNR iteration (src/CGNR.jl:143-178) of the matrix-free normal operator on a dense
column-major ComplexF32 4096x2048 A that is already resident in HBM: t = A p, v = A^H t, then the
fused BLAS-1 update.  lambda = 0, relTol = 0 (SURVEY.md 8d "headline metric run").

N > 1 (launched by torch.distributed.run, one process per GPU): BASELINE config 4 -- independent
solves sharded one per GPU, distinct A per rank, no data-path collective -> weak scaling; the only
collectives are the barrier and the max-over-ranks of the elapsed time.  `--workload rowsharded`
runs BASELINE config 5 instead (one tall A row-partitioned, one all-reduce of A^H t per iteration).

Prints ONE JSON line on rank 0 with `roofline` and (N = 1) `cpu_baseline` objects.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md chip table
# CG converges geometrically on the well-conditioned randn matrix (hard part 4): left running
# for hundreds of iterations the recursive residual underflows Float32 and alpha becomes 0/0.  The
# bench therefore times back-to-back SOLVES of SEGMENT iterations each (BASELINE configs 4/5 use 32);
# the init! of every solve (one extra GEMV) is inside the timed region but not counted as a step.
SEGMENT = 32


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


def cpu_baseline_cgnr(A, b, budget_s=14.0, max_iters=640):
    """the oracle's CGNR (NumPy/OpenBLAS restatement of src/CGNR.jl:143-178) timed on the host.  OpenBLAS's
    cgemv does not scale to every core of a big host, so a short calibration picks the BLAS thread count
    (reported as `cores`) before the bounded timed sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rls_oracle as O

    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_info = threadpool_limits = None

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

    ncpu = os.cpu_count() or 1
    best_threads, calib = ncpu, {}
    if threadpool_limits is not None:
        for nt in sorted({1, 4, 8, 16, 32, ncpu}):
            if nt > ncpu:
                continue
            with threadpool_limits(limits=nt, user_api="blas"):
                run(1, 1.0)  # warm
                n, dt = run(1, 3.0)
            calib[nt] = n / dt
        best_threads = max(calib, key=calib.get)
        ctxmgr = threadpool_limits(limits=best_threads, user_api="blas")
    else:
        import contextlib
        ctxmgr = contextlib.nullcontext()
    with ctxmgr:
        n, dt = run(max_iters // SEGMENT, budget_s)
    return {"value": n / dt, "unit": "iterations/s", "cores": int(best_threads), "kind": "port",
            "sample": f"{n} CGNR iterations of the same {A.shape[0]}x{A.shape[1]} complex64 problem, NumPy/OpenBLAS "
                      f"restatement (oracle/rls_oracle.py), {dt:.1f} s, BLAS threads chosen by calibration "
                      f"{ {k: round(v, 1) for k, v in calib.items()} } it/s of {ncpu} host CPUs",
            "ms_per_step": 1e3 * dt / n}


def other_paths(rls, ctx, Ad, A, b):
    """Untimed extras (N = 1, after the timed region): the other paths of SURVEY 8 on the same operator, a few
    milliseconds each, so that one bench run shows them all.  None of this enters `value`."""
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

    try:
        rho = 0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
        rls.solve_(S, b)
        us = timed(lambda: (rls.init_(S, b), lib.rls_fista_step(S.state._plan, 48)), 48)
        out["fista_l1_matrix_free (BASELINE configs[1])"] = {"us_per_iteration": us, "iterations_per_s": 1e6 / us}
        t0 = time.perf_counter(); G = Ad.gram(); ctx.sync(); t_gram = time.perf_counter() - t0
        t0 = time.perf_counter(); G = Ad.gram(); ctx.sync(); t_gram = min(t_gram, time.perf_counter() - t0)
        out["gram_gemm_AHA (setup, matrix cores)"] = {"ms": 1e3 * t_gram, "TFLOPs_nominal": 8.0 * N * N * M / t_gram / 1e12}
        S = rls.createLinearSolver(rls.CGNR, Ad, AHA=G, iterations=32, relTol=0.0)
        rls.solve_(S, b)
        us = timed(lambda: (rls.init_(S, b), lib.rls_cgnr_step(S.state._plan, 32)), 32)
        out["cgnr_gram_mode (AHA explicit, one launch per iteration)"] = {"us_per_iteration": us, "iterations_per_s": 1e6 / us}
        S = rls.createLinearSolver(rls.FISTA, Ad, AHA=G, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
        rls.solve_(S, b)
        us = timed(lambda: (rls.init_(S, b), lib.rls_fista_step(S.state._plan, 48)), 48)
        out["fista_l1_gram_mode"] = {"us_per_iteration": us, "iterations_per_s": 1e6 / us}
        rng = np.random.default_rng(5)
        for K in (16, 64):
            X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
            Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
            S = rls.createLinearSolver(rls.CGNR, Ad, iterations=32, relTol=0.0)
            rls.solve_(S, Bd, scheduler=rls.BatchedState)
            st = S.state
            us = timed(lambda: (rls._lib.check(h, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 32), "init"),
                                rls._lib.check(h, lib.rls_cgnr_step(st._plan, 32), "step")), 32, reps=4)
            out[f"cgnr_batched_{K}_rhs (BASELINE configs[3] on one GPU, f32 MFMA)"] = {
                "us_per_batched_iteration": us, "solve_iterations_per_s": K * 1e6 / us}
        Xf = (rng.standard_normal((N, 16)) + 1j * rng.standard_normal((N, 16))).astype(np.complex64)
        Bf = rls.DeviceMatrix.from_host(np.asfortranarray((A @ Xf).astype(np.complex64)), ctx)
        S = rls.createLinearSolver(rls.FISTA, Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=48, relTol=0.0)
        rls.solve_(S, Bf, scheduler=rls.BatchedState)
        stf = S.state
        us = timed(lambda: (rls._lib.check(h, lib.rls_fista_init_batched(stf._plan, Bf.ptr, Bf.lda, rho, 1.0, 0.0, 48, 0), "init"),
                            rls._lib.check(h, lib.rls_fista_step(stf._plan, 48), "step")), 48, reps=4)
        out["fista_l1_batched_16_rhs (solve!(FISTA, B), f32 MFMA)"] = {"us_per_batched_iteration": us, "solve_iterations_per_s": 16 * 1e6 / us}
        S = rls.createLinearSolver(rls.Kaczmarz, Ad, reg=rls.L2Regularization(1e-3), iterations=4)
        rls.solve_(S, b); ctx.sync()
        t0 = time.perf_counter(); rls.solve_(S, b); ctx.sync(); dt = time.perf_counter() - t0
        out["kaczmarz_row_sweeps (one launch per solve)"] = {"us_per_row_step": dt / (4 * M) * 1e6}
        # BASELINE configs[2]: ADMM + TV, 8192 x 4096 Float32, shape (64, 64), 10 outer x 10 inner cg! iterations --
        # whole outer iterations enqueued as a device plan (rls_admm_step); wall clock of complete solves, min of 3
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
        out["admm_tv_config3 (BASELINE configs[2], device plan)"] = {
            "ms_per_outer_iteration": ms, "inner_cg_iterations": S.state.cg_iterations,
            "algorithmic_GBps": alg / (ms * 1e-3) / 1e9}
    except Exception as e:  # the extras must never take the headline line down
        out["error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3200)
    ap.add_argument("--warmup", type=int, default=320)
    ap.add_argument("--workload", default="cgnr", choices=["cgnr", "rowsharded"])
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=200)
    args = ap.parse_args()

    import torch  # plumbing: device selection, barrier, max-over-ranks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import rls_amd as rls

    ctx = rls.Context(local_rank)
    M, N = args.M, args.N
    dt = np.complex64
    s = np.dtype(dt).itemsize
    K, W = args.steps, args.warmup

    if args.workload == "rowsharded":
        from importlib import import_module

        mg = import_module("rls_amd.multigpu")
        result = mg.bench_rowsharded(rls, ctx, dist, rank, world, K, W)
        if rank == 0:
            print(json.dumps(result))
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- data: resident in HBM before the timed region ------------------------------------
    A = make_A(M, N, seed=2 if world == 1 else 100 + rank)
    rng = np.random.default_rng(1000 + rank)
    x_true = ((rng.standard_normal(N) + 1j * rng.standard_normal(N)) / math.sqrt(2)).astype(dt)
    b = (A @ x_true).astype(dt)
    Ad = rls.DeviceMatrix.from_host(A, ctx)
    bd = rls.DeviceVector.from_host(b, ctx)
    solver = rls.createLinearSolver(rls.CGNR, Ad, iterations=SEGMENT, relTol=0.0)
    rls.init_(solver, bd)
    st = solver.state
    lib, h = ctx.lib, ctx.handle

    def step(n, initialised=False):
        """n CGNR iterations as back-to-back solves of SEGMENT iterations; `initialised`: the first solve's init! has
        already run (it belongs to the solve's setup, not to its iterations: SURVEY 8d times iterations of a running solve)"""
        while n > 0:
            m = min(n, SEGMENT)
            if not initialised:
                rls.init_(solver, bd)  # asynchronous: r = A^H b, then the init kernel
            initialised = False
            rls._lib.check(h, lib.rls_cgnr_step(st._plan, m), "rls_cgnr_step")
            n -= m

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    # setup (untimed, not part of W): the first LONG host wait of a process (tens of ms of queued GPU
    # work) returns ~50 ms late, once (tools/stall_probe2.py: rep 0 wall 123 ms vs 75 ms of events,
    # every later rep wall == events).  Take that hit here, outside the measurement.
    step(150 * SEGMENT)
    ctx.sync()
    step(W)
    rls.init_(solver, bd)  # the timed region starts on a freshly initialised solve; every later re-init is inside it
    barrier()
    ctx.timer_start()
    t0 = time.perf_counter()
    step(K, initialised=True)
    ev_ms = ctx.timer_stop_ms()  # hipEvents on the stream the kernels run on; synchronises
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st._refresh(lib)
    assert st.iteration == ((K - 1) % SEGMENT) + 1, (st.iteration, K)
    assert math.isfinite(st._residual), "CGNR residual is not finite"

    # ---- per-kernel device time, hipEvents on the ctx stream --------------------------------------
    import ctypes as C

    bytes_iter = bytes_per_cgnr_iteration(M, N, s)
    iter_gbs = bytes_iter * K / (ev_ms * 1e-3) / 1e9
    kern = {}
    us_a, us_r = C.c_float(), C.c_float()
    rls.init_(solver, bd)
    rc = lib.rls_cgnr_step_profiled(st._plan, 10, C.byref(us_a), C.byref(us_r))
    if rc == 0:
        rls.init_(solver, bd)
        rls._lib.check(h, lib.rls_cgnr_step_profiled(st._plan, args.kernel_reps, C.byref(us_a), C.byref(us_r)), "profiled")
        # One launch of the one-pass kernel does BOTH GEMVs of an iteration: its algorithmic bytes are
        # the reference path's 2*M*N*s; what it actually has to move is A once plus the
        # per-workgroup partial rows.
        nwg = M // (2 * 8)
        alg = 2 * M * N * s
        kern["cgnr_pipe_a_kernel (one-pass v = A^H A p + fused CG update)"] = {
            "us_per_launch": us_a.value, "algorithmic_bytes_per_launch": alg, "GBps": alg / (us_a.value * 1e-6) / 1e9,
            "min_hbm_bytes_per_launch": M * N * s + nwg * N * s, "launches_per_iteration": 1}
        rb = nwg * N * s + 3 * N * s
        kern["cgnr_pipe_r_kernel (sum of partial rows + partial dots)"] = {
            "us_per_launch": us_r.value, "algorithmic_bytes_per_launch": rb, "GBps": rb / (us_r.value * 1e-6) / 1e9,
            "launches_per_iteration": 1}
    reps = args.kernel_reps
    p = rls.DeviceVector.from_host(x_true, ctx)
    t = rls.DeviceVector(M, dt, ctx)
    v = rls.DeviceVector(N, dt, ctx)
    for name, fn in (("gemv_n_kernel (two-pass path, t = A p)", lambda: Ad.gemv_(0, p, t)),
                     ("gemv_t_kernel (two-pass path, v = A^H t)", lambda: Ad.gemv_(2, t, v))):
        for _ in range(10):
            fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop_ms() / reps
        by = M * N * s + (M + N) * s
        kern[name] = {"us_per_launch": 1e3 * ms, "algorithmic_bytes_per_launch": by, "GBps": by / (ms * 1e-3) / 1e9,
                      "launches_per_iteration": 0 if rc == 0 else 1}
    used = {k: v_ for k, v_ in kern.items() if v_["launches_per_iteration"] > 0}
    dom = max(used, key=lambda k: used[k]["us_per_launch"])
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    # separate rocprofv3 --pmc runs of tools/pmc_probe.py; tools/pmc_summarize.py); null if not collected
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        traffic = pmc["kernels"][dom.split(" ")[0]]["hbm_bytes_per_launch"]
    except Exception:
        pass

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
            "config": {"workload": f"CGNR matrix-free normal operator, dense column-major ComplexF32 {M}x{N}, lambda=0, "
                                   f"relTol=0, back-to-back solves of {SEGMENT} iterations (BASELINE configs[1] shape; configs[3] sharding at N>1: one independent "
                                   f"solve per GPU, no collectives)",
                       "M": M, "N": N, "problems_per_gpu": 1},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kern[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "achieved = algorithmic bytes (reference path: A read twice per iteration) / device "
                                 "time per launch; the one-pass kernel reads A once, see min_hbm_bytes_per_launch",
                         "per_kernel": kern,
                         "iteration": {"bytes": bytes_iter, "GBps": iter_gbs, "frac": iter_gbs / HBM_PEAK_GBS,
                                       "us_hip_events": 1e3 * ev_ms / K}},
        }
        if world == 1 and not args.no_extras and (M, N) == (4096, 2048):
            out["other_paths"] = other_paths(rls, ctx, Ad, A, bd)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_cgnr(A, b)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
