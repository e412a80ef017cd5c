#!/usr/bin/env python3
"""Gauss-Seidel trip time with option gs_prefix on / off (the residual pass over D + U from the sweep's chain sums against
the pass over A).  usage: gs_prefix_ab.py <lap3d N | cd27 N | pl N | lap2d N> [mc|nat] ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_solvers_amd import capi, matgen, solver as S
TRIPS = int(os.environ.get("TRIPS", "30"))
kind, N = sys.argv[1], int(sys.argv[2])
orders = [{"mc": capi.GS_MULTICOLOR, "nat": capi.GS_LEVEL}[o] for o in (sys.argv[3:] or ["mc"])]
rp, col, val = {"lap3d": matgen.laplace3d_7pt, "cd27": matgen.convdiff3d_27pt, "pl": matgen.powerlaw, "lap2d": matgen.laplace2d_5pt}[kind](N)
n = len(rp) - 1
for order in orders:
    for prefix in (2, 0):
        args = S.Args("gauss-seidel", S.LoopParams(max_iters=10**9, tol=0.0))
        sol = S.Solver("gauss-seidel", device=0, gs_order=order)
        sol.handle.set_option("gs_prefix", prefix)
        t0 = time.perf_counter()
        S.preprocessing_crs(args, sol, rp, col, val, np.ones(n), np.ones(n))
        setup = time.perf_counter() - t0
        h = sol.handle
        for it in range(4): h.iterate(it)
        h.synchronize(); t0 = time.perf_counter()
        for it in range(4, 4 + TRIPS): h.iterate(it)
        h.synchronize()
        print("%s %d order %s gs_prefix %d: %.4f ms/trip (set-up %.2f s, %s)" % (kind, N, "multicolour" if order == capi.GS_MULTICOLOR else "natural", prefix,
              (time.perf_counter() - t0) / TRIPS * 1e3, setup, h.gs_info()), flush=True)
        h.close()
