"""Resident batched Gram kernel against the streaming kernels after the same number of iterations (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rls_amd as rls
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import rls_oracle as O

ctx = rls.default_context(0)
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 2048, 8)))
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
A, X, B = O.make_problem(M, N, np.complex64, 29, n_rhs=K)
B = np.asfortranarray(B)
Ad = rls.DeviceMatrix.from_host(A)
Gd = Ad.gram()
Bd = rls.DeviceMatrix.from_host(B)
out = {}
for res in (0, 2):
    ctx.tune(resident=res)
    S = rls.createLinearSolver(rls.CGNR, Ad, AHA=Gd, iterations=32, relTol=0.0)
    rls.init_(S, Bd, scheduler=rls.BatchedState)
    st = S.state
    st._step(steps)
    stat = st.status()
    out[res] = {n: getattr(st, n).to_host() for n in ("X", "R", "P", "V")}
    out[res]["it"] = [s.iteration for s in stat]
    out[res]["res"] = [s.residual for s in stat]
    out[res]["fb"] = [s.fallbacks for s in stat]
ctx.tune(resident=1)
print("iterations", out[0]["it"], out[2]["it"], "fallbacks", out[2]["fb"])
print("residual  ", np.round(out[0]["res"], 4), np.round(out[2]["res"], 4))
for n in ("X", "R", "P", "V"):
    a, b = out[0][n], out[2][n]
    e = np.linalg.norm(a - b, axis=0) / np.maximum(np.linalg.norm(a, axis=0), 1e-30)
    print(n, "rel diff per column", np.array2string(e, precision=2))
    if n == "V" and steps == 1:
        G = Gd.to_host().astype(np.complex128)
        P0 = (A.conj().T.astype(np.complex128) @ B.astype(np.complex128))
        want = G @ P0
        print("   V vs host G @ P0: streaming", np.linalg.norm(a - want) / np.linalg.norm(want), "resident", np.linalg.norm(b - want) / np.linalg.norm(want))
        ratio = b[:8, :4] / want[:8, :4]
        print("   resident / want, rows 0..7, columns 0..3:\n", np.array2string(ratio, precision=3))
    if e.max() > 1e-4:
        d = np.abs(a - b)
        rows = np.where(d[:, 0] > 1e-4 * np.abs(a[:, 0]).max())[0]
        print("   column 0: wrong rows", len(rows), rows[:16], "...", rows[-4:] if len(rows) else "")
if steps == 1:
    # which parts of the product does the resident V contain?  fit re(V) and im(V) as combinations of the per-wave real products
    G = Gd.to_host().astype(np.complex128)
    P0 = (A.conj().T.astype(np.complex128) @ B.astype(np.complex128))
    comps = []
    names = []
    for wv in range(8):
        sl = slice(256 * wv, 256 * (wv + 1))
        for nm, g_, p_ in (("rr", G.real, P0.real), ("ii", G.imag, P0.imag), ("ri", G.real, P0.imag), ("ir", G.imag, P0.real)):
            comps.append((g_[:, sl] @ p_[sl, :]).ravel())
            names.append(f"w{wv}{nm}")
    Cm = np.stack(comps, axis=1)
    for part, tgt in (("re", out[2]["V"].real.ravel()), ("im", out[2]["V"].imag.ravel())):
        coef, *_ = np.linalg.lstsq(Cm, tgt.astype(np.float64), rcond=None)
        resid = np.linalg.norm(Cm @ coef - tgt) / np.linalg.norm(tgt)
        print(part, "fit residual", f"{resid:.2e}")
        print("  ", " ".join(f"{n}:{c:+.2f}" for n, c in zip(names, coef)))
