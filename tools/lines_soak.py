#!/usr/bin/env python3
"""Soak of the line-pipelined natural-order sweep: many Gauss-Seidel trips, the final iterate and the last recorded norm
against the level schedule's (must be bit-identical).  usage: lines_soak.py <lap3d N | lap2d N> trips [...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
a = sys.argv[1:]
while a:
    kind, N, trips = a[0], int(a[1]), int(a[2]); a = a[3:]
    rp, col, val = {"lap3d": matgen.laplace3d_7pt, "lap2d": matgen.laplace2d_5pt}[kind](N)
    n = len(rp) - 1
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, n); x0 = rng.uniform(-1, 1, n)
    res = {}
    for lines in (1, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=capi.GS_LEVEL)
        sol.handle.set_option("sweep_lines", lines)
        S.preprocessing_crs(args, sol, rp, col, val, b, x0)
        h = sol.handle
        t0 = time.perf_counter()
        for it in range(trips):
            r = h.iterate(it)
        h.synchronize()
        dt = time.perf_counter() - t0
        x = h.download_x(n, trips)
        res[lines] = (r, x)
        print("%s %d, %d trips, sweep_lines %d: %.2f s, last norm %.17g" % (kind, N, trips, lines, dt, r), flush=True)
        h.close()
    same = res[1][0] == res[0][0] and np.array_equal(res[1][1], res[0][1])
    print("   iterates and norms identical: %s" % same, flush=True)
    if not same:
        sys.exit(1)
