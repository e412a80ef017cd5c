#!/usr/bin/env python3
"""Where set-up time goes: C2 (CG, 7-pt 256^3) and C4 (multicolour GS, power-law 10 M rows).
ITS_TIMING=1 makes the library print its own phases on stderr."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ITS_TIMING", "1")
from iterative_solvers_amd import capi, matgen, solver as S

def run(tag, mk, solver, restart=100, **kw):
    rp, col, val = mk()
    n = len(rp) - 1
    args = S.Args(solver, S.LoopParams(max_iters=10**9, tol=0.0, gmres_restart_len=restart))
    sol = S.Solver(solver, device=0, **kw)
    t0 = time.perf_counter()
    S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
    tot = time.perf_counter() - t0
    print(json.dumps({"config": tag, "n": n, "nnz": len(col), "setup_seconds": tot, "python_phases": sol.setup_times}), flush=True)
    sol.handle.close()

which = sys.argv[1:] or ["c2", "c4"]
small = "--small" in which
if "c2" in which:
    run("C2 CG", lambda: matgen.laplace3d_7pt(64 if small else 256), "conjugate-gradient")
if "c3" in which:
    run("C3 GMRES(50)", lambda: matgen.convdiff3d_27pt(24 if small else 128), "gmres", restart=50)
if "c4" in which:
    run("C4 GS multicolour", lambda: matgen.powerlaw(200000 if small else 10_000_000), "gauss-seidel", gs_order=capi.GS_MULTICOLOR)
    run("C4 Jacobi", lambda: matgen.powerlaw(200000 if small else 10_000_000), "jacobi", jacobi_mode=capi.JACOBI_REF)
