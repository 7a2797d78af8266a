#!/usr/bin/env python3
"""OptISTA on the launch-per-iteration path (resident = 0) was seen to drift from 29 to 43-104 us per
iteration within one process, the time going into the host side of rls_optista_update_async.  This probe times, per solve,
(a) the wall clock of the whole enqueue loop, (b) the seconds spent INSIDE each of the two library calls of an iteration, and
(c) the device time of the solve (hipEvents), for OptISTA and POGM, with and without a stream synchronisation between solves;
it prints the series so that a leak (monotone growth of a call's time at constant device time) can be told from queue
back-pressure (call time = device time per launch once the host runs ahead of the device).
usage: python tools/probe_optista_drift.py [solves=120] [iterations=48]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rls_amd as rls  # noqa: E402
from rls_amd import solvers as S  # noqa: E402

n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 120
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 48
M, N = 4096, 2048
rng = np.random.default_rng(2)
A = ((rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))) / np.sqrt(2)).astype(np.complex64)
x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
b = (A @ x).astype(np.complex64)
ctx = rls.default_context(0)
Ad = rls.DeviceMatrix.from_host(np.asfortranarray(A), ctx)
bd = rls.DeviceVector.from_host(b, ctx)
rho = 0.95 / (np.sqrt(M) + np.sqrt(N)) ** 2
lib = ctx.lib


class Timed:
    """wraps a ctypes function: accumulates the seconds spent inside it"""

    def __init__(self, fn):
        self.fn, self.t, self.n = fn, 0.0, 0

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        self.t += time.perf_counter() - t0
        self.n += 1
        return r


class LibProxy:
    def __init__(self, lib, names):
        self._lib = lib
        self.timed = {n: Timed(getattr(lib, n)) for n in names}

    def __getattr__(self, name):
        t = self.__dict__["timed"].get(name)
        return t if t is not None else getattr(self.__dict__["_lib"], name)


for name, upd in (("OptISTA", "rls_optista_update_async"), ("POGM", "rls_pogm_update_async")):
    for sync_between in (True, False):
        ctx.tune(resident=0)
        proxy = LibProxy(lib, ("rls_operator_mul_normal_skip", upd))
        ctx.lib = proxy
        try:
            sol = rls.createLinearSolver(getattr(rls, name), Ad, reg=rls.L1Regularization(1e-2), rho=rho, iterations=iters, relTol=0.0)
            rls.solve_(sol, bd)
            ctx.sync()
            rows = []
            for k in range(n_solves):
                for t in proxy.timed.values():
                    t.t, t.n = 0.0, 0
                rls.init_(sol, bd)
                if sync_between:
                    ctx.sync()
                ctx.timer_start()
                t0 = time.perf_counter()
                sol._run(sol.state)
                wall = time.perf_counter() - t0
                dev_ms = ctx.timer_stop_ms()
                rows.append((1e6 * wall / iters, 1e3 * dev_ms / iters, 1e6 * proxy.timed["rls_operator_mul_normal_skip"].t / iters,
                             1e6 * proxy.timed[upd].t / iters))
        finally:
            ctx.lib = lib
            ctx.tune(resident=1)
        r = np.array(rows)
        pick = [0, 1, 2, 5, 10, 20, 40, 80, n_solves - 1]
        print(f"== {name}, sync between solves = {sync_between}: us per iteration [wall of the enqueue loop + read-back | device (hipEvents) | "
              f"inside mul_normal_skip | inside {upd}]")
        for k in pick:
            if k < len(r):
                print(f"   solve {k:4d}: {r[k, 0]:7.1f} | {r[k, 1]:7.1f} | {r[k, 2]:6.2f} | {r[k, 3]:6.2f}")
        print(f"   first 10 mean {r[:10].mean(0).round(2).tolist()}   last 10 mean {r[-10:].mean(0).round(2).tolist()}")
