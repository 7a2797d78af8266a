"""Per-context tuning state (SURVEY 8b threading row: "all mutable state in rls_ctx; no process-global state";
`solve!` is entered concurrently from several tasks, /root/reference/src/MultiThreading.jl:71): every rls_tune_set switch is a
field of the context it is called on.  Rounds 1-5 kept the per-file measurement switches as file-scope statics."""
import ctypes as C
import os
import re
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rls_oracle as O  # noqa: E402  (the checker)


def test_no_file_scope_tuning_state():
    """`grep -n "^static int g_" csrc/*.hip` is empty (the review's done-criterion), for every integer width"""
    csrc = os.path.join(ROOT, "regularizedleastsquares.jl_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip") or f.endswith(".hpp"):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                if re.match(r"^static (int|int64_t|unsigned|bool|long) g_", line):
                    hits.append(f"{f}:{i}: {line.strip()}")
    assert not hits, hits


@pytest.mark.gpu
def test_two_contexts_two_threads_keep_their_own_switches():
    """Two contexts on one device, one host thread each, different `resident` / `resident_barrier` / `slab_multi` / `skinny_half`
    settings, solving at the same time: each solve reports the kernel path ITS context asked for and matches the float64 oracle;
    flipping a switch on one context never shows on the other."""
    import rls_amd as rls

    M, N = 1024, 2048
    A, xt, b = O.make_problem(M, N, np.complex64, 4242)
    ref = O.CGNR(A.astype(np.complex128), iterations=12, relTol=0.0)
    O.solve(ref, b.astype(np.complex128))
    settings = [dict(resident=1, resident_barrier=2, slab_multi=1), dict(resident=0, resident_barrier=1, slab_multi=0)]
    want_path = [4, 1]  # resident launch | two-launch slab pipeline (rls_cgnr_path)
    out = [None, None]
    errs = []

    go = threading.Barrier(2)

    def work(i):
        try:
            ctx = rls.Context(0)
            ctx.tune(**settings[i])
            Ad = rls.DeviceMatrix.from_host(A, ctx)
            bd = rls.DeviceVector.from_host(b, ctx)
            # one solve before the two threads meet: on a cold box the first launch of a kernel loads its code object (tens of
            # milliseconds), long enough for the OTHER thread's resident launch to give up waiting for the chip and take its fallback --
            # correct, but not what this test is about (it is the FIRST test the GPU tier runs, on a fresh machine)
            rls.solve_(rls.createLinearSolver(rls.CGNR, Ad, iterations=12, relTol=0.0), bd)
            ctx.sync()
            go.wait(timeout=120)
            paths, xs, lost = [], [], 0
            for rep in range(6):
                S = rls.createLinearSolver(rls.CGNR, Ad, iterations=12, relTol=0.0)
                x = rls.solve_(S, bd).to_host()
                p = C.c_int32(-1)
                ctx.lib.rls_cgnr_path(S.state._plan, C.byref(p))
                paths.append(p.value)
                xs.append(x)
                lost += int(S.state._refresh(ctx.lib).fallbacks)
            out[i] = (paths, xs, lost)
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
            try:
                go.abort()
            except Exception:  # noqa: BLE001
                pass

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for i in range(2):
        paths, xs, lost = out[i]
        for x in xs:
            assert np.linalg.norm(x - ref.x) / np.linalg.norm(ref.x) < 1e-5
        if lost == 0:   # (a resident launch that could not get the chip next to the other thread's kernels re-runs on the pipeline: other bits, path 1 next)
            assert paths == [want_path[i]] * 6, (i, paths)
            assert all(np.array_equal(x, xs[0]) for x in xs)  # run to run, the same bits on one context
        else:
            assert i == 0 and set(paths) <= {4, 1}, (i, paths, lost)
    assert out[1][0] == [1] * 6   # the context that switched the resident kernels off never ran one

    # the batched layout switch and the TV limits are the context's, too: what one context sets, another does not see
    c1, c2 = rls.Context(0), rls.Context(0)
    c1.tune(skinny_half=0, tv_fused_max_n=16, tv_fused_2d=0)
    img = np.random.default_rng(3).standard_normal(64 * 8).astype(np.float32)
    want = O.prox_tv_fgp(img.copy().astype(np.float64), 0.2, (64, 8), None, 10)
    for c in (c1, c2):
        got = rls.prox_(rls.TVRegularization, rls.DeviceVector.from_host(img, c), 0.2, shape=(64, 8)).to_host()
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-5
    # ... observable through the FISTA plan: context 1's limits refuse the single-workgroup FGP launch (the solver falls back to the
    # primitives), context 2's accept it; both match the oracle
    At, _, bt = O.make_problem(3 * 512, 512, np.float32, 99)
    rho = 0.9 / np.linalg.norm(At.astype(np.float64), 2) ** 2
    o = O.FISTA(At.astype(np.float64), reg=O.TVRegularization(0.05, shape=(64, 8)), rho=rho, iterations=8, relTol=0.0)
    xo = np.array(O.solve(o, bt.astype(np.float64)))
    for c, planned in ((c1, False), (c2, True)):
        s_ = rls.createLinearSolver(rls.FISTA, rls.DeviceMatrix.from_host(At, c), reg=rls.TVRegularization(0.05, shape=(64, 8)), rho=rho,
                                    iterations=8, relTol=0.0)
        x = rls.solve_(s_, rls.DeviceVector.from_host(bt, c)).to_host()
        assert bool(s_.state._plan) == planned
        assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-5


COLD = r"""
import sys, threading
sys.path.insert(0, sys.argv[1])
import numpy as np
import rls_amd as rls
rng = np.random.default_rng(1)
M, N = 1024, 2048
A = np.asfortranarray((rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))).astype(np.complex64))
b = (A @ (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)).astype(np.complex64)
out, errs = [None, None], []
start = threading.Barrier(2)
def work(i):
    try:
        ctx = rls.Context(0)
        ctx.tune(resident=1 - i)          # thread 0: the resident kernel, thread 1: the two-launch pipeline -- both need > 64 KiB of LDS
        Ad, bd = rls.DeviceMatrix.from_host(A, ctx), rls.DeviceVector.from_host(b, ctx)
        S = rls.createLinearSolver(rls.CGNR, Ad, iterations=10, relTol=0.0)
        start.wait()                      # the FIRST launches of the process, from two threads at once
        out[i] = rls.solve_(S, bd).to_host()
    except Exception as e:
        errs.append((i, repr(e)))
ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errs, errs
assert np.all(np.isfinite(out[0])) and np.linalg.norm(out[0] - out[1]) < 1e-4 * np.linalg.norm(out[0])
print("ok")
"""


@pytest.mark.gpu
def test_first_launches_of_a_cold_process_from_two_threads():
    """The one-time work behind a kernel's first launch (its > 64 KiB dynamic-LDS attribute, `rls_device_once`) is complete before ANY
    thread launches: two threads entering their first solves at the same time in a fresh process (src/MultiThreading.jl:71).  The
    second thread used to see "done" while the first was still setting attributes -- a cold-process failure, one run in a few."""
    import subprocess

    pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for _ in range(4):
        r = subprocess.run([sys.executable, "-c", COLD, pkg], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
