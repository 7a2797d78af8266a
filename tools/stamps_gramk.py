"""diagnostic: phase timeline of one iteration of cgnr_gramk_resident_kernel -- or, with the argument `fista`, of
fista_gramk_resident_kernel -- (needs the -DRLS_STAMPS build, tools/build_stamps.sh -> tools/ubench/librls_stamps.so).  Stamps are those of the LAST iteration of the launch; also
prints the launch time per iteration for several step counts (the slope is the iteration, the intercept the one-off
load of AHA + the gather of x)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rls_amd as rls
import rls_amd._lib as L
_variant = next((a[4:] for a in sys.argv if a.startswith("lib=")), "stamps")
if os.path.exists(os.path.join(ROOT, "tools", "ubench", f"librls_{_variant}.so")) and "nostamps" not in sys.argv:
    L.LIB_PATH = os.path.join(ROOT, "tools", "ubench", f"librls_{_variant}.so")
    L._lib = None
from bench import make_A
ctx = rls.Context(0)
lib = ctx.lib
M, N, K = 4096, 2048, 8
A = make_A(M, N, 4); Ad = rls.DeviceMatrix.from_host(A, ctx)
rng = np.random.default_rng(5)
X = (rng.standard_normal((N, K)) + 1j * rng.standard_normal((N, K))).astype(np.complex64)
Bd = rls.DeviceMatrix.from_host(np.asfortranarray((A @ X).astype(np.complex64)), ctx)
if "fista" in sys.argv:
    rho = float(0.9 / np.linalg.norm(A.astype(np.complex128), 2) ** 2)
    S = rls.createLinearSolver(rls.FISTA, Ad, AHA=Ad.gram(), reg=rls.L1Regularization(1e-2), rho=rho, iterations=2000, relTol=0.0)
    rls.init_(S, Bd, scheduler=rls.BatchedState)
    st = S.state
    init = lambda: rls._lib.check(ctx.handle, lib.rls_fista_init_batched(st._plan, Bd.ptr, Bd.lda, rho, 1.0, 0.0, 2000, 0), "init")
    for n in (4, 8, 16, 32, 64):
        init(); st._step(n); ctx.sync()
        ts = []
        for _ in range(5):
            init(); ctx.sync()
            ctx.timer_start(); st._step(n); ts.append(ctx.timer_stop_ms() * 1e3)
        print(f"step({n:2d}): {np.median(ts):8.2f} us per call = {np.median(ts)/n:6.2f} us per iteration")
    if hasattr(lib, "rls_debug_gk_stamps"):
        init(); st._step(20); ctx.sync()
        buf = (C.c_ulonglong * 128)()
        lib.rls_debug_gk_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
        print("status", lib.rls_debug_gk_stamps(buf))
        names = ["iteration start", "products done (wave partials in LDS)", "own rows updated, rows of y + partial norms published, drained",
                 "grid barrier passed", "panel gathered into LDS, partial norms summed per thread", "retirement flags known"]
        t00 = min(buf[wg * 16] for wg in range(7))
        for wg in range(7):
            t = [buf[wg * 16 + i] for i in range(6)]
            print(f"wg {wg*37+5}: start @{(t[0]-t00)*10:+5d} ns  " + "  ".join(f"[{i}] +{(t[i]-t[0])*10}" for i in range(1, 6)))
        print("phases: " + "; ".join(f"[{i}] {n}" for i, n in enumerate(names)))
        t = [buf[7 * 16 + i] for i in range(16)]
        print(f"workgroup 0, whole launch of 20 iterations (us): kernel start -> state loaded (AHA rows, y, own rows) {(t[9]-t[8])/100:.1f}; iterations "
              f"{(t[10]-t[9])/100:.1f}; rows published + final barrier {(t[11]-t[10])/100:.1f}; workgroup 0 writes the caller's state {(t[12]-t[11])/100:.1f} (x / xold / res gathered and stored {(t[13]-t[11])/100:.1f}, y + operand panel {(t[14]-t[13])/100:.1f})")
    sys.exit(0)
S = rls.createLinearSolver(rls.CGNR, Ad, AHA=Ad.gram(), iterations=2000, relTol=0.0)
rls.init_(S, Bd, scheduler=rls.BatchedState)
st = S.state
for n in (4, 8, 16, 32, 64):
    rls.init_(S, Bd, scheduler=rls.BatchedState); st = S.state
    st._step(n); ctx.sync()
    ts = []
    for _ in range(5):
        rls._lib.check(ctx.handle, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 2000), "init"); ctx.sync()
        ctx.timer_start(); st._step(n); ts.append(ctx.timer_stop_ms() * 1e3)
    print(f"step({n:2d}): {np.median(ts):8.2f} us per call = {np.median(ts)/n:6.2f} us per iteration")
if hasattr(lib, "rls_debug_gk_stamps"):
    rls._lib.check(ctx.handle, lib.rls_cgnr_init_batched(st._plan, Bd.ptr, Bd.lda, 0.0, 0.0, 2000), "init")
    st._step(20); ctx.sync()
    buf = (C.c_ulonglong * 128)()
    lib.rls_debug_gk_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    print("status", lib.rls_debug_gk_stamps(buf))
    names = ["iteration start", "products done (wave partials in LDS)", "rows of V + partial dots published, drained",
             "grid barrier passed", "dots summed, alpha known", "r updated (V read), local ||r||^2", "beta known", "p updated in the panel"]
    t00 = min(buf[wg * 16] for wg in range(7))
    for wg in range(7):
        t = [buf[wg * 16 + i] for i in range(8)]
        print(f"wg {wg*37+5}: start @{(t[0]-t00)*10:+5d} ns  " + "  ".join(f"[{i}] +{(t[i]-t[0])*10}" for i in range(1, 8)))
    print("phases: " + "; ".join(f"[{i}] {n}" for i, n in enumerate(names)))
    t = [buf[7 * 16 + i] for i in range(16)]
    print(f"workgroup 0, whole launch of 20 iterations (us): kernel start -> state loaded (AHA rows, r, p) {(t[9]-t[8])/100:.1f}; iterations "
          f"{(t[10]-t[9])/100:.1f}; x published + final barrier {(t[11]-t[10])/100:.1f}; workgroup 0 writes the caller's state {(t[12]-t[11])/100:.1f}")
