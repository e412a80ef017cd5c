#!/usr/bin/env python3
"""Jacobi / Gauss-Seidel trip time on the 256^3 7-pt Laplacian for the exact-SpMV variants."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
N = int(os.environ.get("N", "256"))
rp, col, val = matgen.laplace3d_7pt(N)
n = len(rp) - 1
for name in sys.argv[1:] or ["jacobi"]:
    for variant in (4, 5, 2):
        args = S.Args(name, S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver(name, device=0, jacobi_mode=capi.JACOBI_REF)
        sol.handle.set_option("spmv_variant", variant)
        sol.handle.set_option("spmv_grid", 1024 if variant == 2 else 512)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        h = sol.handle
        for it in range(5): h.iterate(it)
        h.synchronize(); t0 = time.perf_counter()
        for it in range(5, 55): h.iterate(it)
        h.synchronize()
        print("%s variant %d: %.4f ms/trip" % (name, variant, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
        h.close()
