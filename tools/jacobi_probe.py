#!/usr/bin/env python3
"""Jacobi trip time on the 7-pt Laplacian (LAP=N), shipped (MODE=0) or reference (MODE=1) sweep, FUSED=0/1."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
N = int(os.environ.get("LAP", "256"))
rp, col, val = matgen.laplace3d_7pt(N)
n = len(rp) - 1
for mode in (0, 1):
    for fused in (0, 1):
        args = S.Args("jacobi", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("jacobi", device=0, jacobi_mode=mode)
        sol.handle.set_option("fused_residual", fused)
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        h = sol.handle
        for it in range(5): h.iterate(it)
        h.synchronize(); t0 = time.perf_counter()
        for it in range(5, 55): h.iterate(it)
        h.synchronize()
        print("mode %d (%s) fused %d: %.4f ms/trip" % (mode, "shipped" if mode == 0 else "reference", fused, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
        h.close()
